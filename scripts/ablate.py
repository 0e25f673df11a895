"""Timing-only ablations of the 16-bit scan (instrumented build, MMF_SCAN_DEBUG bits; results are WRONG by design):
64 = instrumented build, nothing removed; +1 no filter / list code; +2 no tile DMA; +4 no barrier."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
from bench import make_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512          # scripts/ablate.py 131072 1024 half: the split-k variant
X = make_rows(0, N, D, torch.device("cuda", 0))
if len(sys.argv) > 3 and sys.argv[3] == "half":
    X = X.half()
modes = [("production", None), ("dbg build", 64), ("no list code", 65), ("no DMA", 66), ("no barrier", 68), ("no list, no DMA", 67),
         ("no list, no barrier", 69), ("no list, DMA, barrier", 71), ("no list, DMA of tile 0 only", 64 + 1 + 32)]
res = {m: [] for m, _ in modes}
for r in range(4):
    for name, bits in modes:
        if bits is None:
            os.environ.pop("MMF_SCAN_DEBUG", None)
        else:
            os.environ["MMF_SCAN_DEBUG"] = str(bits)
        try:
            _, _, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", profile=True, return_stats=True)
            if r:
                res[name].append(st["scan_ms"])
        except RuntimeError as e:
            print(name, "failed:", str(e)[:100])
os.environ.pop("MMF_SCAN_DEBUG", None)
for name, _ in modes:
    if res[name]:
        t = statistics.median(res[name])
        print(f"{name:28s} {t:8.3f} ms  {2.0 * N * N * D / t / 1e9 / 2500.0:.3f} of peak")
