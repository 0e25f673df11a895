"""The oracle against the golden vectors captured from the reference itself (CPU only).

Layer 1 (oracle.ref_restate, torch-CPU, reference op order) must reproduce the fixtures to float
rounding; layer 2 (oracle.mmf_oracle.c, canonical fmaf chains) must agree with them to 1e-5 on
scores and exactly on indices for the tie-free fixtures.  This is what "parity pinned" rests on.
"""
import json

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_restate as rr
from conftest import load_golden

TOL = 1e-5


def T(a):
    return torch.from_numpy(np.asarray(a))


# ---------------------------------------------------------------- G1: dense a1 / a2 / a3
@pytest.mark.parametrize("N,D", [(2, 4), (64, 32), (256, 128)])
def test_g1_dense(N, D):
    g = load_golden("g1_dense.npz")
    X, P2, P3 = g[f"N{N}_D{D}_X"], g[f"N{N}_D{D}_P2"], g[f"N{N}_D{D}_P3"]
    lams = (0.5, 1.0, 2.0) if N <= 64 else (1.0,)
    for lam in lams:
        Kh = g[f"N{N}_D{D}_lam{lam}_Kh"]
        # layer 1: same torch ops -> bitwise on the same machine/library, float rounding elsewhere
        np.testing.assert_allclose(rr.compute_morphological_similarity(T(X), lam).numpy(), Kh, rtol=0, atol=2e-6)
        np.testing.assert_allclose(rr.compute_spatial_similarity(T(P2), lam).numpy(), g[f"N{N}_D{D}_lam{lam}_Kg2"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(rr.compute_combined_similarity(T(X), T(P3), lam, 2.0 * lam).numpy(),
                                   g[f"N{N}_D{D}_lam{lam}_K3"], rtol=0, atol=2e-6)
        # layer 2: canonical chain
        np.testing.assert_allclose(oracle.sim_dense(X, metric="rbf", lam=lam), Kh, rtol=0, atol=TOL)
        np.testing.assert_allclose(oracle.sim_dense(P2, metric="rbf", lam=lam), g[f"N{N}_D{D}_lam{lam}_Kg2"], rtol=0, atol=TOL)
        np.testing.assert_allclose(oracle.sim_dense(P3, metric="rbf", lam=lam), g[f"N{N}_D{D}_lam{lam}_Kg3"], rtol=0, atol=TOL)
        np.testing.assert_allclose(oracle.sim_dense_combined(X, P2, lam, lam), g[f"N{N}_D{D}_lam{lam}_K2"], rtol=0, atol=TOL)
        np.testing.assert_allclose(oracle.sim_dense_combined(X, P3, lam, 2.0 * lam), g[f"N{N}_D{D}_lam{lam}_K3"], rtol=0, atol=TOL)
    # dummy zero positions (preprocess_hypergraph.py:59) -> K_g == 1 -> K == K_h
    np.testing.assert_allclose(oracle.sim_dense_combined(X, np.zeros((N, 2), np.float32), 1.0, 1.0),
                               g[f"N{N}_D{D}_Kzero"], rtol=0, atol=TOL)


# ---------------------------------------------------------------- G2: a4 / a6 threshold builder
@pytest.mark.parametrize("N", [2, 8, 64])
@pytest.mark.parametrize("ratio", [0.0, 0.5, 1.0, 2.0])
def test_g2_threshold(N, ratio):
    g = load_golden("g2_threshold.npz")
    X, P = g[f"N{N}_X"], g[f"N{N}_P"]
    ei_ref, ew_ref = g[f"N{N}_r{ratio}_ei"], g[f"N{N}_r{ratio}_ew"]
    ei1, ew1 = rr.build_weighted_hypergraph(T(X), T(P), 1.0, 1.0, ratio)
    assert np.array_equal(ei1.numpy(), ei_ref)
    np.testing.assert_allclose(ew1.numpy(), ew_ref, rtol=0, atol=2e-6)
    # canonical layer: same median element and the same edge set unless an entry sits within
    # rounding of the threshold (none does on these fixtures)
    K = oracle.sim_dense_combined(X, P, 1.0, 1.0)
    med = oracle.offdiag_lower_median(K)
    ei2, ew2 = oracle.threshold_edges(K, med * ratio)
    assert np.array_equal(ei2, ei_ref)
    np.testing.assert_allclose(ew2, ew_ref, rtol=0, atol=TOL)


def test_g2_errors_and_data():
    g = load_golden("g2_threshold.npz")
    errs = json.loads(str(g["errors_json"]))
    assert errs == {"ratio_none": "TypeError", "n1": "ValueError"}   # SURVEY.md Appendix A1, similarity_kernel.py:176
    with pytest.raises(TypeError):
        rr.build_weighted_hypergraph(torch.randn(4, 8), torch.rand(4, 2), 1.0, 1.0, None)
    with pytest.raises(ValueError):
        rr.build_weighted_hypergraph(torch.randn(1, 8), torch.rand(1, 2), 1.0, 1.0, 0.5)
    with pytest.raises(ValueError):
        oracle.offdiag_lower_median(np.ones((1, 1), np.float32))
    for N in (2, 8, 64):
        np.testing.assert_allclose(rr.mean_pool_with_similarity(T(g[f"N{N}_X"])).numpy(), g[f"N{N}_data_pool"], atol=1e-7)


def test_lower_median_even_count():
    K = np.array([[9, 1, 2], [3, 9, 4], [5, 6, 9]], dtype=np.float32)   # off-diagonal 1..6 -> lower median 3
    assert oracle.offdiag_lower_median(K) == 3.0
    assert torch.median(torch.tensor([1., 2., 3., 4., 5., 6.])).item() == 3.0


# ---------------------------------------------------------------- G3: a7 cross-modal
@pytest.mark.parametrize("N,M", [(8, 12), (100, 37)])
def test_g3_cross(N, M):
    g = load_golden("g3_cross.npz")
    A, B = g[f"N{N}_M{M}_A"], g[f"N{N}_M{M}_B"]
    for lam in ("1", "0.5"):
        S_ref = g[f"N{N}_M{M}_S_lam{lam}"]
        S1, st = rr.compute_wsi_tma_similarity(T(A), None, T(B), float(lam))
        np.testing.assert_allclose(S1.numpy(), S_ref, rtol=0, atol=2e-6)
        np.testing.assert_allclose([st[k] for k in ("mean", "std", "min", "max", "median")],
                                   g[f"N{N}_M{M}_stats_lam{lam}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(oracle.sim_dense(A, B, metric="rbf_direct", lam=float(lam)), S_ref, rtol=0, atol=TOL)
        # the norm-expansion form agrees with the direct form to tolerance on unit-norm rows
        np.testing.assert_allclose(oracle.sim_dense(A, B, metric="rbf", lam=float(lam)), S_ref, rtol=0, atol=TOL)


# ---------------------------------------------------------------- G4: a8 sklearn kNN, tie-free
@pytest.mark.parametrize("N,D", [(64, 32), (512, 128)])
@pytest.mark.parametrize("k", [1, 5, 16])
def test_g4_knn(N, D, k):
    g = load_golden("g4_knn.npz")
    X = g[f"N{N}_D{D}_X"]
    ind = g[f"N{N}_D{D}_k{k}_ind"]       # [N, k+1], column 0 is self on tie-free data
    dist = g[f"N{N}_D{D}_k{k}_dist"]
    assert np.array_equal(ind[:, 0], np.arange(N))
    # layer 1
    pairs = rr.knn_pairs_exact(T(X), k)
    assert np.array_equal(pairs[:, 1].reshape(N, k), ind[:, 1:])
    # layer 2, every metric that is rank-equivalent on unit-norm rows
    for metric in ("neg_sq_l2", "cosine", "rbf", "dot"):
        idx, val = oracle.simtopk(X, metric=metric, lam=1.0, k=k)
        assert np.array_equal(idx, ind[:, 1:]), metric
    idx, val = oracle.simtopk(X, metric="neg_sq_l2", k=k)
    np.testing.assert_allclose(-val, dist[:, 1:] ** 2, rtol=0, atol=TOL)


# ---------------------------------------------------------------- G5: whole build_hypergraph_knn_kmeans
@pytest.mark.parametrize("tag", ["small", "zero", "mid"])
def test_g5_knn_kmeans(tag):
    g = load_golden("g5_knn_kmeans.npz")
    W, Tm = g[f"{tag}_W"], g[f"{tag}_T"]
    k, H = int(g[f"{tag}_k"]), int(g[f"{tag}_H"])
    allf = np.concatenate([W, Tm], 0)
    labels = g[f"{tag}_labels"]
    ei_ref, ew_ref = g[f"{tag}_ei_sorted"], g[f"{tag}_ew_sorted"]
    assert ei_ref.shape[1] == int(g[f"{tag}_num_edges"])
    # canonical kNN (self dropped by identity) + cliques from the recorded labels + dedup + weights
    idx, _ = oracle.simtopk(allf, metric="neg_sq_l2", k=k)
    n = allf.shape[0]
    knn = np.stack([np.repeat(np.arange(n), k), idx.reshape(-1)], 1)
    pairs = np.concatenate([knn, rr.clique_pairs(labels, H)], 0)
    ei1, ew1 = rr.dedup_and_weight(T(allf), pairs)
    if tag != "zero":
        assert np.array_equal(ei1.numpy(), ei_ref)
        np.testing.assert_allclose(ew1.numpy(), ew_ref, rtol=0, atol=2e-6)
        np.testing.assert_allclose(oracle.edge_cosine(allf, ei_ref), ew_ref, rtol=0, atol=TOL)
    else:
        # the zero row ties every distance from it at ||x_j||^2 = 1: sklearn's pick among those is
        # unspecified; every other row must match, and so must every weight.
        keep_ref = ~((ei_ref[0] == 3) | (ei_ref[1] == 3))
        e1 = ei1.numpy()
        keep_1 = ~((e1[0] == 3) | (e1[1] == 3))
        assert np.array_equal(e1[:, keep_1], ei_ref[:, keep_ref])
        w = oracle.edge_cosine(allf, ei_ref)
        np.testing.assert_allclose(w, ew_ref, rtol=0, atol=TOL)
        assert np.all(w[~keep_ref] == 0.0)          # cosine with a zero row is 0 (eps clamp), :419


# ---------------------------------------------------------------- G6: ties
def test_g6_ties_dup_and_lattice():
    g = load_golden("g6_ties.npz")
    for tag in ("dup", "lattice"):
        X, ind, dist = g[f"{tag}_X"], g[f"{tag}_ind"], g[f"{tag}_dist"]
        N = X.shape[0]
        # sklearn's order among ties is unspecified: compare the distance MULTISET of its k+1 picks
        # (self included) with the canonical k picks + self
        idx, val = oracle.simtopk(X, metric="neg_sq_l2", k=5)
        ours = np.sort(np.concatenate([np.zeros((N, 1)), np.sqrt(np.maximum(-val.astype(np.float64), 0))], 1), 1)
        np.testing.assert_allclose(ours, np.sort(dist, 1), rtol=0, atol=2e-3 if tag == "dup" else 1e-6)
        # canonical tie-break: equal keys come out in ascending id order, never self
        assert not np.any(idx == np.arange(N)[:, None])
        for r in range(N):
            for t in range(4):
                if val[r, t] == val[r, t + 1]:
                    assert idx[r, t] < idx[r, t + 1]


# ---------------------------------------------------------------- G7: a5
def test_g7_pool():
    g = load_golden("g7_pool.npz")
    np.testing.assert_allclose(rr.mean_pool_with_similarity(T(g["X"])).numpy(), g["pool1"], atol=1e-7)
    np.testing.assert_allclose(rr.mean_pool_with_similarity(T(g["X"]), T(g["P"]), 1.0, 1.0).numpy(), g["pool2"], atol=1e-7)


# ---------------------------------------------------------------- oracle self-consistency
def test_simtopk_equals_topk_of_dense():
    X = np.random.RandomState(3).randn(200, 48).astype(np.float32)
    Y = np.random.RandomState(4).randn(333, 48).astype(np.float32)
    for metric in ("dot", "cosine", "neg_sq_l2", "rbf"):
        lam = 0.01 if metric == "rbf" else 1.0
        D = oracle.sim_dense(X, Y, metric=metric, lam=lam)
        idx, val = oracle.simtopk(X, Y, metric=metric, lam=lam, k=7)
        assert np.array_equal(np.take_along_axis(D, idx, 1), val)
        if metric != "rbf":   # rbf ranks by the exponent; expf may merge neighbours
            order = np.lexsort((np.broadcast_to(np.arange(333), D.shape), -D), axis=1)[:, :7]
            assert np.array_equal(order, idx)


def test_offsets_and_sharding_equivalence():
    X = np.random.RandomState(5).randn(96, 32).astype(np.float32)
    full_i, full_v = oracle.simtopk(X, metric="cosine", k=4)
    parts_i, parts_v = [], []
    for r in range(3):
        i, v = oracle.simtopk(X[32 * r:32 * (r + 1)], X, metric="cosine", k=4, exclude_self=True, row_offset=32 * r)
        parts_i.append(i)
        parts_v.append(v)
    assert np.array_equal(np.concatenate(parts_i), full_i)
    assert np.array_equal(np.concatenate(parts_v), full_v)
    # column panels + merge
    a = oracle.simtopk(X, X[:40], metric="cosine", k=4, exclude_self=True)
    b = oracle.simtopk(X, X[40:], metric="cosine", k=4, exclude_self=True, col_offset=40)
    mi, mv = oracle.topk_merge(a[0], a[1], b[0], b[1])
    assert np.array_equal(mi, full_i) and np.array_equal(mv, full_v)


def test_bad_arguments():
    X = np.zeros((4, 8), np.float32)
    with pytest.raises(ValueError):
        oracle.simtopk(X, k=4)                       # k > N-1 admissible columns
    with pytest.raises(ValueError):
        oracle.simtopk(X, metric="rbf", lam=0.0, k=1)
    oracle.simtopk(X, k=3)
