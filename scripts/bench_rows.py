"""Roofline numbers for the other rows of SURVEY.md §8a (everything except the N x N scan that bench.py measures):
each kernel's algorithmic bytes / flops per call (DESIGN.md §4.2–4.4) divided by its measured time, against the HBM
roof (8 TB/s) or the f32-MFMA roof (157.3 TFLOP/s).  Prints one JSON object; profiles/r01_rows.json is a copy."""
import json, sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
ops = mmf.ops
dev = torch.device('cuda')
HBM, F32 = 8000.0, 157.3


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def rows(n, d, seed, unit=True):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((n, d), generator=g, device=dev)
    return x / x.norm(dim=1, keepdim=True) if unit else x


out = {}
# a1-a3: dense similarity matrices at the reference's scale (N ~ 1e4): HBM-write-bound, N^2 * 4 B
N, d = 16384, 512
F, P = rows(N, d, 1, False) * 0.05, torch.rand((N, 2), device=dev) * 100
t = timed(lambda: ops.sim_dense(F, metric="rbf", lam=0.5))
out["a1 sim_dense rbf N=16384 d=512"] = {"ms": t * 1e3, "bound": "f32 mfma", "TFLOP/s": 2 * N * N * d / t / 1e12, "frac": 2 * N * N * d / t / 1e12 / F32,
                                          "write GB/s": N * N * 4 / t / 1e9}
t = timed(lambda: ops.sim_dense(P, metric="rbf", lam=0.001))
out["a2 sim_dense rbf N=16384 d=2 (spatial)"] = {"ms": t * 1e3, "bound": "hbm", "GB/s": N * N * 4 / t / 1e9, "frac": N * N * 4 / t / 1e9 / HBM}
t = timed(lambda: ops.sim_dense_combined(F, P, 0.5, 0.001))
out["a3 sim_dense_combined N=16384 d=512+2"] = {"ms": t * 1e3, "bound": "f32 mfma", "TFLOP/s": 2 * N * N * d / t / 1e12, "frac": 2 * N * N * d / t / 1e12 / F32,
                                                "write GB/s": N * N * 4 / t / 1e9}
# a4-a5: median + threshold edges on a materialised K
K = ops.sim_dense_combined(F, P, 0.5, 0.001)
t = timed(lambda: ops.offdiag_lower_median(K))
out["a4 offdiag_lower_median N=16384"] = {"ms": t * 1e3, "bound": "hbm", "GB/s": 4 * N * N * 4 / t / 1e9, "frac": 4 * N * N * 4 / t / 1e9 / HBM,
                                          "note": "4 radix passes over K"}
thr = float(ops.offdiag_lower_median(K))
ei, ew = ops.threshold_edges(K, thr)
E = ei.shape[1]
t = timed(lambda: ops.threshold_edges(K, thr), reps=5)
out["a5 threshold_edges N=16384"] = {"ms": t * 1e3, "edges": E, "bound": "hbm", "GB/s": (2 * N * N * 4 + E * 20) / t / 1e9,
                                     "frac": (2 * N * N * 4 + E * 20) / t / 1e9 / HBM, "note": "count pass + fill pass over K, 20 B written per edge"}
del K, ei, ew
# a7: edge weights of a k-NN graph
N2, k = 262144, 5
X = rows(N2, d, 3)
idx, _ = mmf.simtopk(X, metric="cosine", k=k)
ei = torch.stack([torch.arange(N2, device=dev).repeat_interleave(k), idx.reshape(-1)])
t = timed(lambda: ops.edge_cosine(X, ei))
out["a7 edge_cosine N=262144 k=5 d=512"] = {"ms": t * 1e3, "bound": "hbm gather", "GB/s": ei.shape[1] * 2 * d * 4 / t / 1e9,
                                            "frac": ei.shape[1] * 2 * d * 4 / t / 1e9 / HBM}
# exact f32 scan (precision="exact"): the AUTO path for d > 1024 or k + self > 12, and the rescan of flagged rows
N3 = 65536
X3 = rows(N3, d, 4)
t = timed(lambda: mmf.simtopk(X3, metric="cosine", k=5, precision="exact"), reps=3, warm=1)
out["a8 exact f32 scan N=65536 d=512"] = {"ms": t * 1e3, "bound": "f32 mfma", "TFLOP/s": 2 * N3 * N3 * d / t / 1e12, "frac": 2 * N3 * N3 * d / t / 1e12 / F32}
# small pieces of the fast path
t = timed(lambda: ops.row_scalars(X, "cosine", torch.empty(N2, device=dev)))
out["row_scalars N=262144 d=512"] = {"ms": t * 1e3, "bound": "hbm", "GB/s": N2 * d * 4 / t / 1e9, "frac": N2 * d * 4 / t / 1e9 / HBM}
print(json.dumps(out, indent=1))
