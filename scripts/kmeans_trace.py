"""Where the device KMeans spends its time: iterations, host wall per step (run plain, or under rocprofv3 --kernel-trace)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
from importlib import import_module
km = import_module("multimodal_fusion_amd.kmeans")
ops = mmf.ops
rng = np.random.RandomState(0)
for N, D, S in ((16384, 512, 100), (65536, 512, 100)):
    cent = rng.randn(200, D).astype(np.float32)
    W = torch.from_numpy((cent[rng.randint(0, 200, N)] * 0.3 + 0.05 * rng.randn(N, D)).astype(np.float32)).cuda()
    km.kmeans_fit_predict(W[:2048], 8, n_init=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lab, C, inertia = km.kmeans_fit_predict(W, S)
    torch.cuda.synchronize(); total = (time.perf_counter() - t0) * 1e3
    # pieces, timed alone
    Xc = W - W.mean(0, keepdim=True)
    def t(fn, reps=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    Cc = C - W.mean(0, keepdim=True)
    gen = torch.Generator(device="cuda").manual_seed(1)
    print(f"N={N} D={D} S={S}: fit {total:7.1f} ms | assign {t(lambda: km._assign(Xc, Cc)):6.3f} ms  segment_sort {t(lambda: ops.segment_sort(lab, S)):6.3f}  "
          f"segment_mean {t(lambda: ops.segment_mean(Xc, ops.segment_sort(lab, S))):6.3f}  kmeans++ (10 seedings) {t(lambda: km._kmeanspp(Xc, S, 10, gen), 3):7.2f} ms", flush=True)
