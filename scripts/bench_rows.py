"""Roofline numbers for the rows of SURVEY.md §8 other than the N x N top-k scan that bench.py measures: each kernel's
algorithmic bytes / flops per call divided by its measured time, against the roof that bounds it:
HBM 8 TB/s, f32 MFMA 157.3 TFLOP/s, or — for the direct-difference RBF — the vector ALU (2 instructions per pair and k:
3.93e13 pair-k/s at 2.4 GHz).  Prints one JSON object; profiles/rNN_rows.json is a copy.  Keys start with the §8 row id."""
import json, os, sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
ops = mmf.ops
dev = torch.device('cuda')
HBM, F32, VALU_PAIRK = 8000.0, 157.3, 3.93e13


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


def hbm(t, nbytes, **extra):
    return dict({"ms": t * 1e3, "bound": "hbm", "GB/s": nbytes / t / 1e9, "frac": nbytes / t / 1e9 / HBM}, **extra)


def mfma(t, flop, **extra):
    return dict({"ms": t * 1e3, "bound": "f32 mfma", "TFLOP/s": flop / t / 1e12, "frac": flop / t / 1e12 / F32}, **extra)


out = {}
# a1-a3: dense similarity matrices (compute_*_similarity) at N = 16384
N, d = 16384, 512
F, P = rows(N, d, 1, False) * 0.05, torch.rand((N, 2), device=dev) * 100
t = timed(lambda: ops.sim_dense(F, metric="rbf", lam=0.5))
out["a1 compute_morphological_similarity: sim_dense rbf N=16384 d=512"] = mfma(t, 2 * N * N * d, **{"write GB/s": N * N * 4 / t / 1e9})
t = timed(lambda: ops.sim_dense(P, metric="rbf", lam=0.001))
out["a2 compute_spatial_similarity: sim_dense rbf N=16384 d=2"] = hbm(t, N * N * 4, note="N^2 * 4 B written")
t = timed(lambda: ops.sim_dense_combined(F, P, 0.5, 0.001))
out["a3 compute_combined_similarity: sim_dense_combined N=16384 d=512+2"] = mfma(t, 2 * N * N * d, **{"write GB/s": N * N * 4 / t / 1e9})
# a4: build_weighted_hypergraph = lower median of the off-diagonal + ordered threshold compaction on a materialised K
K = ops.sim_dense_combined(F, P, 0.5, 0.001)
t = timed(lambda: ops.offdiag_lower_median(K))
out["a4 build_weighted_hypergraph: offdiag_lower_median N=16384"] = hbm(t, N * N * 4, note="one sweep over K (sampled bracket + exact counts) + a select among the 3 % inside the bracket")
os.environ["MMF_MEDIAN_RADIX"] = "1"
t = timed(lambda: ops.offdiag_lower_median(K))
del os.environ["MMF_MEDIAN_RADIX"]
out["a4 build_weighted_hypergraph: offdiag_lower_median N=16384, four-pass radix select (MMF_MEDIAN_RADIX=1)"] = hbm(t, 4 * N * N * 4, note="4 radix passes over K")
thr = float(ops.offdiag_lower_median(K))
ei, ew = ops.threshold_edges(K, thr)
E = ei.shape[1]
t = timed(lambda: ops.threshold_edges(K, thr), reps=5)
out["a4 build_weighted_hypergraph: threshold_edges N=16384"] = hbm(t, 2 * N * N * 4 + E * 20, edges=E, note="count pass + fill pass over K, 20 B written per edge")
# f4: the five statistics of a stored matrix (aggregate_wsi_super_patches' K_wsi): one reduction pass + 4 radix passes
t = timed(lambda: ops.array_stats(K), reps=5)
out["f4 similarity statistics of a stored matrix: array_stats N=16384"] = hbm(t, 2 * N * N * 4, note="1 reduction pass + 1 median sweep; torch: 4 reductions + a sort")
# f2: the same median with K never stored (recomputed in row panels)
for env in ("", "1"):
    if env:
        os.environ["MMF_MEDIAN_RADIX"] = env
    t = timed(lambda: ops.combined_offdiag_median(F, P, 0.5, 0.001), reps=3, warm=1)
    os.environ.pop("MMF_MEDIAN_RADIX", None)
    out["f2 combined_offdiag_median N=16384 d=512, K never stored" + (", four-pass radix select" if env else "")] = mfma(
        t, (4 if env else 1) * 2 * N * N * d, note=("4" if env else "1") + " recomputation(s) of K")
del K, ei, ew
# a7: compute_wsi_tma_similarity — direct-difference RBF (VALU bound) with the statistics fused
M = 16384
A, B = rows(N, d, 5) * 0.7, rows(M, d, 6) * 0.7
for name, fn in (("matrix only", lambda: ops.sim_dense(A, B, metric="rbf_direct", lam=1.0)),
                 ("matrix + mean/std/min/max/median", lambda: ops.sim_dense_stats(A, B, metric="rbf_direct", lam=1.0)),
                 ("statistics only, matrix never stored (one recomputation: the median's single sweep)", lambda: ops.sim_dense_stats(A, B, metric="rbf_direct", lam=1.0, store=False))):
    t = timed(fn, reps=3, warm=1)
    work = N * M * d
    out[f"a7 compute_wsi_tma_similarity: rbf_direct {N}x{M} d={d}, {name}"] = {
        "ms": t * 1e3, "bound": "valu (v_sub + v_fma per pair-k)", "pair-k/s": work / t, "frac": work / t / VALU_PAIRK,
        "write GB/s": 0.0 if "never" in name else N * M * 4 / t / 1e9}
del A, B
# a8 (exact): the f32 MFMA scan — precision="exact", the AUTO path for d > 1024 (d > 512 when k + self > 20), and the rescan of flagged rows
N3 = 65536
X3 = rows(N3, d, 4)
t = timed(lambda: mmf.simtopk(X3, metric="cosine", k=5, precision="exact"), reps=3, warm=1)
out["a8 k-NN, exact f32 scan N=65536 d=512 k=5"] = mfma(t, 2 * N3 * N3 * d)
t = timed(lambda: mmf.simtopk(X3, metric="cosine", k=32, precision="exact"), reps=3, warm=1)
out["a8 k-NN, exact f32 scan N=65536 d=512 k=32 (48-entry lists)"] = mfma(t, 2 * N3 * N3 * d)
del X3
# a9: edge weights of a k-NN graph
N2, k = 262144, 5
X = rows(N2, d, 3)
idx, _ = mmf.simtopk(X, metric="cosine", k=k)
ei = torch.stack([torch.arange(N2, device=dev).repeat_interleave(k), idx.reshape(-1)])
t = timed(lambda: ops.edge_cosine(X, ei))
out["a9 edge weights: edge_cosine N=262144 k=5 d=512"] = dict(hbm(t, ei.shape[1] * 2 * d * 4), bound="hbm gather")
t = timed(lambda: ops.knn_pairs(idx))
out["a9 undirected dedup: knn_pairs N=262144 k=5"] = dict(hbm(t, N2 * k * (8 + 16)), bound="hbm gather", note="8 B read + 16 B written per pair (+ k reverse look-ups)")
# a10 / f3: cluster-shaped steps on 100 super patches of 262144 patches, and the cliques of 10 hyperedges of 20000 nodes
lab = torch.randint(0, 100, (N2,), device=dev)
t = timed(lambda: ops.segment_sort(lab, 100))
out["a10 members of each cluster: segment_sort N=262144 S=100"] = hbm(t, N2 * 24, note="labels read twice, order written")
seg = ops.segment_sort(lab, 100)
t = timed(lambda: ops.segment_mean(X, seg))
out["a10 mean pooling: segment_mean N=262144 d=512 S=100"] = dict(hbm(t, N2 * d * 4), bound="hbm gather")
seg2 = ops.segment_sort(torch.randint(0, 10, (20000,), device=dev), 10)
lo, hi = ops.clique_pairs(seg2)
t = timed(lambda: ops.clique_pairs(seg2), reps=5)
out["a10 clique expansion: clique_pairs N=20000 H=10"] = hbm(t, lo.numel() * 16, pairs=lo.numel(), note="16 B written per pair (count launch + fill launch)")
# small pieces of the fast path
t = timed(lambda: ops.row_scalars(X, "cosine", torch.empty(N2, device=dev)))
out["row_scalars N=262144 d=512"] = hbm(t, N2 * d * 4)
print(json.dumps(out, indent=1))
