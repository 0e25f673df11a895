"""Where the device KMeans (mmf_kmeans_fit) spends its time.  Plain: wall time per fit; under
`rocprofv3 --kernel-trace --stats -- python3 scripts/kmeans_trace.py`: the per-kernel table."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import multimodal_fusion_amd as mmf
from importlib import import_module
from make_g9_kmeans import g9_data
km = import_module("multimodal_fusion_amd.kmeans")
reps = int(os.environ.get("REPS", "5"))
cases = [("g9 clustered", g9_data("clustered"), 100), ("g9 gauss", g9_data("gauss"), 100)]
rng = np.random.RandomState(0)
cent = rng.randn(200, 512).astype(np.float32)
cases.append(("clustered N=65536", (cent[rng.randint(0, 200, 65536)] * 0.3 + 0.05 * rng.randn(65536, 512)).astype(np.float32), 100))
S = np.exp(-0.05 * ((rng.randn(100, 1, 16) - rng.randn(1, 256, 16)) ** 2).sum(-1)).astype(np.float32)
cases.append(("similarity rows 100 x 256, 10 groups", S, 10))
if os.environ.get("ONLY"):
    cases = [c for c in cases if os.environ["ONLY"] in c[0]]
for tag, X, k in cases:
    Xg = torch.from_numpy(X).cuda()
    n = X.shape[0]
    first, u = km.sklearn_stream(42, 10, k, n)
    mmf.ops.kmeans_fit(Xg, k, first, u)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        lab, C, info = mmf.ops.kmeans_fit(Xg, k, first, u)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        km.sklearn_stream(42, 10, k, n)
    host = (time.perf_counter() - t0) / reps * 1e3
    its = info["n_iter_per_init"]
    fma = n * 10 * k * X.shape[1] * info["lockstep_iterations"]
    print(f"{tag:40s} fit {ms:7.2f} ms (+ {host:.2f} ms host random stream)  lockstep iterations {info['lockstep_iterations']}  per restart {its}  "
          f"E-step multiply-adds {fma:.3g}", flush=True)
