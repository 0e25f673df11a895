"""d <= 1024: the split-k pair kernel against the one-wave-per-SIMD kernel (MMF_SCAN_NO_SPLITK=1), same process."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
from bench import make_rows
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 1024
X = make_rows(0, N, d, torch.device("cuda", 0)).half()
res = {"splitk": [], "nw4": []}
ref = None
for r in range(5):
    for name in ("splitk", "nw4"):
        if name == "nw4":
            os.environ["MMF_SCAN_NO_SPLITK"] = "1"
        else:
            os.environ.pop("MMF_SCAN_NO_SPLITK", None)
        i, v, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", profile=True, return_stats=True)
        if ref is None:
            ref = (i, v)
        assert torch.equal(i, ref[0]) and torch.equal(v, ref[1]), name
        if r:
            res[name].append(st["scan_ms"])
os.environ.pop("MMF_SCAN_NO_SPLITK", None)
for name, t in res.items():
    m = statistics.median(t)
    print(f"{name:8s} scan {m:8.3f} ms  frac {2.0 * N * N * d / (m * 1e-3) / 2.5e15:.4f}  fallback {st['fallback_rows']}")
