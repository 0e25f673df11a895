"""Timing-only ablations of the exact f32 kernels (MMF_F32_DEBUG: 1 no epilogue, 2 no staging), same process.
Results with a bit set are wrong by construction; only the times mean something."""
import os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device("cuda", 0)
X = make_rows(0, 65536, 512, dev)
F = make_rows(0, 16384, 512, dev) * 0.3
P = torch.rand((16384, 2), device=dev)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


res = {}
for r in range(5):
    for bits in (0, 1, 2, 3):
        os.environ["MMF_F32_DEBUG"] = str(bits)
        def scan():
            try:
                mmf.simtopk(X, metric="cosine", k=5, precision="exact")
            except RuntimeError:       # the invariant check trips on the ablated kernel's lists — after everything has run
                pass
        a = timed(scan)
        b = timed(lambda: mmf.ops.sim_dense(F, metric="rbf", lam=0.5))
        c = timed(lambda: mmf.ops.sim_dense_combined(F, P, 0.5, 1.0))
        if r:
            res.setdefault(bits, []).append((a, b, c))
os.environ["MMF_F32_DEBUG"] = "0"
fl_s, fl_d = 2 * 65536 ** 2 * 512, 2 * 16384 ** 2 * 512
for bits, name in ((0, "whole kernel"), (1, "no epilogue"), (2, "no staging"), (3, "neither: LDS reads + MFMA + barrier")):
    a, b, c = (statistics.median(x[i] for x in res[bits]) for i in range(3))
    print(f"{name:40s} exact call {a:7.2f} ms ({fl_s / a / 1e9 / 157.3:.3f})  sim_dense {b:6.3f} ms ({fl_d / b / 1e9 / 157.3:.3f})  combined {c:6.3f} ms ({fl_d / c / 1e9 / 157.3:.3f})")
