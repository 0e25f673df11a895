"""In-process A/B of the exact f32 kernels between library variants: scripts/ab_f32.py base f32old"""
import os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_fusion_amd as mmf
from bench import make_rows
names = sys.argv[1:]
dev = torch.device("cuda", 0)
X = make_rows(0, 65536, 512, dev)
F = make_rows(0, 16384, 512, dev) * 0.3
P = torch.rand((16384, 2), device=dev)
libs, res = {}, {}
def use(name):
    path = os.path.join(ROOT, "multimodal-fusion_amd", "libmmf_hg.so" if name == "base" else f"libmmf_hg_{name}.so")
    mmf._lib._lib = libs.get(name); mmf._lib.SO_PATH = path
    if libs.get(name) is None:
        libs[name] = mmf._lib.lib()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for r in range(6):
    for n in names:
        use(n)
        a = timed(lambda: mmf.simtopk(X, metric="cosine", k=5, precision="exact"))
        b = timed(lambda: mmf.ops.sim_dense(F, metric="rbf", lam=0.5))
        c = timed(lambda: mmf.ops.sim_dense_combined(F, P, 0.5, 1.0))
        if r:
            res.setdefault(n, []).append((a, b, c))
for n in names:
    print(n, "exact scan %.2f ms  sim_dense %.3f ms  combined %.3f ms" % tuple(statistics.median(x[i] for x in res[n]) for i in range(3)))
