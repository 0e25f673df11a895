import os, sys, torch
sys.path.insert(0, os.getcwd())
import multimodal_fusion_amd as mmf
from bench import make_rows
for N in (65536, 131072):
    X = make_rows(0, N, 512, torch.device("cuda", 0))
    for name, bits in (("production", None), ("instrumented", 64), ("counters", 72), ("no list code", 65), ("no barrier", 68), ("no DMA", 66)):
        if bits is None: os.environ.pop("MMF_SCAN_DEBUG", None)
        else: os.environ["MMF_SCAN_DEBUG"] = str(bits)
        for rep in range(3):
            try:
                _, _, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", profile=True, return_stats=True)
            except RuntimeError as e:
                pass
        print(f"N={N} {name:14s} scan {st['scan_ms']:7.3f} ms  frac {2.0*N*N*512/(st['scan_ms']*1e-3)/2.5e15:.3f}", flush=True)
    os.environ.pop("MMF_SCAN_DEBUG", None)
