"""The 16-bit scan on Gaussian and clustered rows: event counters (MMF_SCAN_DEBUG=72), and timing-only ablations of the
instrumented build — without the list code (65), without the overflow-list stores (192: results wrong by design)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
from bench import make_rows
for data in ("gaussian", "clustered"):
    X = make_rows(0, 262144, 512, torch.device("cuda", 0), data=data)
    for name, bits in (("production", None), ("instrumented", 64), ("counters", 72), ("no overflow-list stores", 192), ("no list code", 65)):
        if bits is None:
            os.environ.pop("MMF_SCAN_DEBUG", None)
        else:
            os.environ["MMF_SCAN_DEBUG"] = str(bits)
        ms = float("nan")
        for rep in range(2):
            try:
                _, _, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", profile=True, return_stats=True)
                ms = st["scan_ms"]
            except RuntimeError as e:
                print("  (", str(e)[:90], ")")
        print(f"{data:10s} {name:24s} scan {ms:8.2f} ms", flush=True)
    os.environ.pop("MMF_SCAN_DEBUG", None)
