"""In-process A/B of the direct-difference RBF kernel between library variants: scripts/ab_direct.py prev base"""
import os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_fusion_amd as mmf
from bench import make_rows
names = sys.argv[1:]
dev = torch.device("cuda", 0)
A = make_rows(0, 16384, 512, dev) * 0.7
B = make_rows(16384, 32768, 512, dev) * 0.7
libs, res, ref = {}, {}, None
def use(name):
    path = os.path.join(ROOT, "multimodal-fusion_amd", "libmmf_hg.so" if name == "base" else f"libmmf_hg_{name}.so")
    mmf._lib._lib = libs.get(name); mmf._lib.SO_PATH = path
    if libs.get(name) is None:
        libs[name] = mmf._lib.lib()
for r in range(6):
    for n in names:
        use(n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S = mmf.ops.sim_dense(A, B, metric="rbf_direct", lam=1.0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        if ref is None:
            ref = S
        assert torch.equal(S, ref), n
        if r:
            res.setdefault(n, []).append(dt)
for n in names:
    m = statistics.median(res[n])
    print(f"{n:8s} {m:7.3f} ms  {16384 * 16384 * 512 / (m * 1e-3) / 3.93e13:.3f} of the VALU roof")
