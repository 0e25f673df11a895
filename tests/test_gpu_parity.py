"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): top-k indices bit-exact under the stated tie-break (key desc, global
column id asc, self dropped by identity); scores within 1e-5 (they are in fact bitwise for every
metric but rbf, whose expf differs from libm by ulps).
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, unit_rows

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def mmf():
    import multimodal_fusion_amd as m
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return m


def dev(a):
    return torch.as_tensor(np.asarray(a)).cuda()


def rnd(n, d, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(n, d) * scale).astype(np.float32)


# ------------------------------------------------------------------ canonical chain == f32 MFMA
@pytest.mark.parametrize("n,m,d", [(7, 5, 3), (130, 257, 32), (64, 64, 100), (200, 300, 512), (129, 1, 513)])
def test_dense_dot_is_the_canonical_chain_bitwise(mmf, n, m, d):
    X, Y = rnd(n, d, 1), rnd(m, d, 2)
    got = mmf.sim_dense(dev(X), dev(Y), metric="dot").cpu().numpy()
    ref = oracle.sim_dense(X, Y, metric="dot")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("metric", ["cosine", "neg_sq_l2", "rbf", "rbf_direct"])
def test_dense_metrics(mmf, metric):
    X, Y = rnd(150, 64, 3, 0.2), rnd(90, 64, 4, 0.2)
    got = mmf.sim_dense(dev(X), dev(Y), metric=metric, lam=0.7).cpu().numpy()
    ref = oracle.sim_dense(X, Y, metric=metric, lam=0.7)
    if metric in ("cosine", "neg_sq_l2"):
        assert np.array_equal(got, ref)
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)
    got = mmf.sim_dense(dev(X), metric=metric, lam=0.7).cpu().numpy()
    np.testing.assert_allclose(got, oracle.sim_dense(X, metric=metric, lam=0.7), rtol=0, atol=TOL)


# ------------------------------------------------------------------ fused similarity + top-k
def check_topk(mmf, X, Y, metric, k, lam=1.0, precision="exact", **kw):
    idx, val = mmf.simtopk(dev(X), None if Y is None else dev(Y), metric=metric, lam=lam, k=k, precision=precision, **kw)
    ridx, rval = oracle.simtopk(X, Y, metric=metric, lam=lam, k=k, **kw)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert np.array_equal(idx, ridx), f"{metric} k={k}: {(idx != ridx).sum()} index mismatches"
    if metric == "rbf":
        np.testing.assert_allclose(val, rval, rtol=0, atol=TOL)
    else:
        assert np.array_equal(val, rval)


@pytest.mark.parametrize("metric", ["dot", "cosine", "neg_sq_l2", "rbf"])
@pytest.mark.parametrize("n,d,k", [(2, 4, 1), (33, 3, 5), (300, 32, 5), (1000, 128, 16), (257, 100, 27), (2049, 512, 5)])
def test_simtopk_self_exact(mmf, metric, n, d, k):
    X = unit_rows(n, d, 10 + n).numpy() if d > 3 else rnd(n, d, 5)
    check_topk(mmf, X, None, metric, k, lam=1.0)


@pytest.mark.parametrize("metric", ["cosine", "neg_sq_l2"])
@pytest.mark.parametrize("n,d,k,exclude", [(400, 64, 28, True), (400, 64, 32, True), (3000, 128, 32, True), (3000, 128, 43, True),
                                           (700, 96, 44, False), (5000, 256, 33, True)])
def test_simtopk_large_k(mmf, metric, n, d, k, exclude):
    """sklearn's NearestNeighbors has no cap on n_neighbors (preprocess_hypergraph.py:379); the exact scan's 48-entry
    lists carry k + self up to 44, and so do the 16-bit scan's 32-entry list pairs for d <= 512 (AUTO's choice)."""
    X = unit_rows(n, d, 100 + n + k).numpy()
    if exclude:
        check_topk(mmf, X, None, metric, k, precision="exact")
        check_topk(mmf, X, None, metric, k, precision="auto")
    else:
        Y = unit_rows(n + 50, d, 7).numpy()
        check_topk(mmf, X, Y, metric, k, precision="auto")


@pytest.mark.parametrize("metric", ["dot", "cosine", "neg_sq_l2", "rbf"])
@pytest.mark.parametrize("n,d,k,exclude", [(400, 64, 45, True), (1000, 128, 64, True), (3000, 512, 100, True), (700, 96, 44 * 3, False),
                                           (600, 1024, 87, True)])
def test_simtopk_k_beyond_one_pass(mmf, metric, n, d, k, exclude):
    """k + self > 44 (round 3): several passes of the exact scan, each offering only the columns that rank strictly after the
    previous pass's last entry in the total order (key desc, id asc).  scikit-learn's n_neighbors has no cap (:379).  Against the
    oracle: indices bit-exact, scores bitwise (rbf to 1e-5); duplicate rows make ties that straddle the pass boundaries."""
    X = unit_rows(n, d, 500 + n + k).numpy()
    X[50:90] = X[50]                                # 40 exact copies: equal keys, ids decide — also across a pass boundary
    if exclude:
        check_topk(mmf, X, None, metric, k, precision="auto")
    else:
        Y = unit_rows(n + 50, d, 9).numpy()
        Y[10:70] = Y[10]
        check_topk(mmf, X, Y, metric, k, precision="auto", exclude_self=False)
    with pytest.raises(RuntimeError, match="does not support"):
        mmf.simtopk(dev(X), metric=metric, k=k, precision="fast")


@pytest.mark.parametrize("metric", ["dot", "cosine", "neg_sq_l2", "rbf"])
@pytest.mark.parametrize("n,d,k", [(300, 32, 20), (1000, 128, 21), (3000, 128, 32), (2049, 512, 27), (5000, 256, 43), (777, 500, 40)])
def test_simtopk_large_k_fast_lists(mmf, metric, n, d, k):
    """k + self in 21..44 on the 16-bit scan: 32-entry lane lists with 5 slot bits (d <= 512)."""
    X = unit_rows(n, d, 300 + n + k).numpy()
    for prec in ("fast", "fast_bf16"):
        check_topk(mmf, X, None, metric, k, precision=prec)
    _, _, st = mmf.simtopk(dev(X), metric=metric, k=k, return_stats=True)
    assert st["precision_used"] == 2, st          # AUTO takes it too


def test_simtopk_large_k_fast_splits_offsets_duplicates(mmf):
    X = unit_rows(4096, 128, 6).numpy()
    full_i, full_v = oracle.simtopk(X, metric="cosine", k=30)
    for splits in (1, 2, 8, 16):
        i, v, st = mmf.simtopk(dev(X), metric="cosine", k=30, precision="fast", col_splits=splits, return_stats=True)
        assert np.array_equal(i.cpu().numpy(), full_i) and np.array_equal(v.cpu().numpy(), full_v), splits
        assert st["fallback_rows"] == 0, (splits, st)
    i, v = mmf.simtopk(dev(X[1000:1777]), dev(X), metric="cosine", k=30, exclude_self=True, row_offset=1000, precision="fast")
    assert np.array_equal(i.cpu().numpy(), full_i[1000:1777]) and np.array_equal(v.cpu().numpy(), full_v[1000:1777])
    # cross-modal, no self: k + self = k
    Y = unit_rows(3000, 128, 8).numpy()
    check_topk(mmf, X[:900], Y, "neg_sq_l2", 44, precision="fast", exclude_self=False)
    # 60 exact copies of 10 rows: ties fill the lists, the overflow lists take the rest, ids break the ties
    D = np.repeat(rnd(10, 64, 3), 60, axis=0)
    idx, val, st = mmf.simtopk(dev(D), metric="neg_sq_l2", k=25, precision="fast", return_stats=True)
    ridx, rval = oracle.simtopk(D, metric="neg_sq_l2", k=25)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
    assert st["precision_used"] == 2
    # a tight cluster inside a random set
    X = unit_rows(3000, 128, 77).numpy()
    X[100:200] = X[100] + 1e-4 * rnd(100, 128, 78)
    ridx, rval = oracle.simtopk(X, metric="cosine", k=24)
    for splits in (1, 0):
        idx, val, st = mmf.simtopk(dev(X), metric="cosine", k=24, precision="fast", col_splits=splits, return_stats=True)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
    # d > 512 keeps large k on the exact scan; asking for the 16-bit scan there is an error
    Z = unit_rows(600, 700, 9).numpy()
    _, _, st = mmf.simtopk(dev(Z), metric="cosine", k=30, return_stats=True)
    assert st["precision_used"] == 1
    with pytest.raises(RuntimeError, match="does not support"):
        mmf.simtopk(dev(Z), metric="cosine", k=30, precision="fast")


# the bf16 MFMA scan + exact re-rank must give the SAME bits as the exact scan and the oracle
@pytest.mark.parametrize("metric", ["dot", "cosine", "neg_sq_l2", "rbf"])
@pytest.mark.parametrize("n,d,k", [(2, 4, 1), (33, 3, 5), (300, 32, 5), (1000, 128, 7), (257, 100, 7), (2049, 512, 5),
                                   (5000, 256, 5), (777, 500, 3), (3000, 128, 11), (1500, 512, 9), (900, 1000, 10)])
def test_simtopk_self_fast(mmf, metric, n, d, k):
    X = unit_rows(n, d, 10 + n).numpy() if d > 3 else rnd(n, d, 5)
    check_topk(mmf, X, None, metric, k, lam=1.0, precision="fast")
    check_topk(mmf, X, None, metric, k, lam=1.0, precision="fast_bf16")


def test_simtopk_fast_extreme_scales(mmf):
    # the exact power-of-two scale of the f16 operands must cope with tiny and huge row norms
    for scale in (1e-20, 1e-6, 1e4, 1e15):
        X = rnd(700, 96, 41) * np.float32(scale)
        for metric in ("dot", "cosine", "neg_sq_l2"):
            check_topk(mmf, X, None, metric, 4, precision="fast")
    X = rnd(600, 64, 42)
    X[::7] *= 1e-4            # mixed norms: small rows get a wide margin, must still be exact
    X[5] = 0.0
    for metric in ("dot", "cosine", "neg_sq_l2"):
        check_topk(mmf, X, None, metric, 4, precision="fast")


def test_simtopk_fast_unnormalised_and_rect(mmf):
    X, Y = rnd(900, 200, 31, 3.0), rnd(2100, 200, 32, 0.05)      # very different row norms
    for metric in ("dot", "cosine", "neg_sq_l2"):
        check_topk(mmf, X, Y, metric, 6, precision="fast")
        check_topk(mmf, Y, X, metric, 4, precision="fast")
    check_topk(mmf, X * 0.02, None, "rbf", 5, lam=0.5, precision="fast")


def test_simtopk_fast_overflow_falls_back_to_exact(mmf):
    # 40 exact copies of 10 rows: far more columns sit inside the f16 margin than a lane list can hold; they go to
    # the rows' overflow lists (192 slots) and the exact re-rank sorts them out — no row needs the exact rescan
    D = np.repeat(rnd(10, 64, 3), 40, axis=0)
    idx, val, st = mmf.simtopk(dev(D), metric="neg_sq_l2", k=5, precision="fast", return_stats=True)
    ridx, rval = oracle.simtopk(D, metric="neg_sq_l2", k=5)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
    assert st["precision_used"] == 2 and st["fallback_rows"] == 0 and st["candidates"] >= 400 * 39
    # 300 copies of 4 rows: more than the overflow lists hold, so the rows are flagged and rescanned by the exact
    # kernel; the answer must not change
    D = np.repeat(rnd(4, 64, 4), 300, axis=0)
    idx, val, st = mmf.simtopk(dev(D), metric="neg_sq_l2", k=5, precision="fast", return_stats=True)
    ridx, rval = oracle.simtopk(D, metric="neg_sq_l2", k=5)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
    assert st["precision_used"] == 2 and st["fallback_rows"] == 1200 and st["overflow_rows"] == st["fallback_rows"]
    # a tight cluster inside an otherwise random set: the cluster's rows take their neighbours from the overflow lists
    X = unit_rows(3000, 128, 77).numpy()
    X[100:160] = X[100] + 1e-4 * rnd(60, 128, 78)
    ridx, rval = oracle.simtopk(X, metric="cosine", k=5)
    for splits in (1, 0):
        idx, val, st = mmf.simtopk(dev(X), metric="cosine", k=5, precision="fast", col_splits=splits, return_stats=True)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
        assert st["fallback_rows"] == 0, st
    # ... and a cluster larger than the overflow lists: only ITS rows are rescanned
    X = unit_rows(3000, 128, 79).numpy()
    X[100:400] = X[100] + 1e-4 * rnd(300, 128, 80)
    ridx, rval = oracle.simtopk(X, metric="cosine", k=5)
    idx, val, st = mmf.simtopk(dev(X), metric="cosine", k=5, precision="fast", col_splits=1, return_stats=True)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
    assert 300 <= st["fallback_rows"] <= 600, st     # + rows whose own top-k touches the cluster


def test_simtopk_fast_forced_splits_and_offsets(mmf):
    X = unit_rows(4096, 128, 5).numpy()
    full_i, full_v = oracle.simtopk(X, metric="cosine", k=5)
    for splits in (1, 2, 8, 16):
        i, v = mmf.simtopk(dev(X), metric="cosine", k=5, precision="fast", col_splits=splits)
        assert np.array_equal(i.cpu().numpy(), full_i) and np.array_equal(v.cpu().numpy(), full_v), splits
    i, v = mmf.simtopk(dev(X[1000:1777]), dev(X), metric="cosine", k=5, exclude_self=True, row_offset=1000, precision="fast")
    assert np.array_equal(i.cpu().numpy(), full_i[1000:1777]) and np.array_equal(v.cpu().numpy(), full_v[1000:1777])


@pytest.mark.parametrize("metric", ["cosine", "neg_sq_l2"])
def test_simtopk_rect_exact(mmf, metric):
    X, Y = rnd(333, 96, 7), rnd(1500, 96, 8)
    check_topk(mmf, X, Y, metric, 7)
    check_topk(mmf, Y, X, metric, 3)


def test_simtopk_offsets_and_forced_splits(mmf):
    X = rnd(640, 64, 11)
    full_i, full_v = oracle.simtopk(X, metric="cosine", k=6)
    parts = []
    for r in range(5):
        i, v = mmf.simtopk(dev(X[128 * r:128 * (r + 1)]), dev(X), metric="cosine", k=6, exclude_self=True,
                           row_offset=128 * r, precision="exact", col_splits=1 + r)
        parts.append((i.cpu().numpy(), v.cpu().numpy()))
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full_i)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full_v)
    a = mmf.simtopk(dev(X), dev(X[:200]), metric="cosine", k=6, exclude_self=True, precision="exact")
    b = mmf.simtopk(dev(X), dev(X[200:]), metric="cosine", k=6, exclude_self=True, col_offset=200, precision="exact")
    mi, mv = mmf.topk_merge(a[0], a[1], b[0], b[1])
    assert np.array_equal(mi.cpu().numpy(), full_i) and np.array_equal(mv.cpu().numpy(), full_v)


def test_simtopk_ties_and_duplicates(mmf):
    g = load_golden("g6_ties.npz")
    for tag in ("dup", "lattice"):
        X = g[f"{tag}_X"]
        check_topk(mmf, X, None, "neg_sq_l2", 5)
        check_topk(mmf, X, None, "cosine", 5)
    Z = np.zeros((70, 16), np.float32)          # every key ties: ids come out ascending, self skipped
    check_topk(mmf, Z, None, "dot", 9)
    D = np.repeat(rnd(10, 32, 3), 40, axis=0)   # 40 copies of 10 rows
    check_topk(mmf, D, None, "neg_sq_l2", 12)


def test_simtopk_half_inputs(mmf):
    X = unit_rows(500, 64, 21)
    for dt in (torch.bfloat16, torch.float16):
        Xh = X.to(dt)
        ridx, rval = oracle.simtopk(Xh.float().numpy(), metric="cosine", k=5)
        for prec in ("exact", "fast"):
            idx, val = mmf.simtopk(Xh.cuda(), metric="cosine", k=5, precision=prec)
            assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval), (dt, prec)


def test_golden_knn_through_gpu(mmf):
    g = load_golden("g4_knn.npz")
    for (N, D) in [(64, 32), (512, 128)]:
        X = g[f"N{N}_D{D}_X"]
        for k in (1, 5, 16):
            idx, val = mmf.simtopk(dev(X), metric="neg_sq_l2", k=k)
            assert np.array_equal(idx.cpu().numpy(), g[f"N{N}_D{D}_k{k}_ind"][:, 1:])   # sklearn's own answer
            np.testing.assert_allclose(-val.cpu().numpy(), g[f"N{N}_D{D}_k{k}_dist"][:, 1:] ** 2, rtol=0, atol=TOL)


def test_errors(mmf):
    X = dev(rnd(4, 8, 1))
    with pytest.raises(ValueError):
        mmf.simtopk(X, k=4)                 # only 3 admissible columns
    with pytest.raises(ValueError):
        mmf.simtopk(X, k=0)
    with pytest.raises(ValueError):
        mmf.simtopk(X, metric="rbf", lam=0.0, k=1)
    with pytest.raises(RuntimeError):
        mmf.simtopk(X.cpu(), k=1)           # no CPU path
    mmf.simtopk(X, k=3)


# ------------------------------------------------------------------ list kernels
def test_edge_cosine_and_merge(mmf):
    X = rnd(300, 48, 9)
    X[17] = 0.0
    ei = np.random.RandomState(1).randint(0, 300, size=(2, 5000)).astype(np.int64)
    got = mmf.edge_cosine(dev(X), dev(ei)).cpu().numpy()
    assert np.array_equal(got, oracle.edge_cosine(X, ei))
    g = load_golden("g5_knn_kmeans.npz")
    allf = np.concatenate([g["mid_W"], g["mid_T"]], 0)
    got = mmf.edge_cosine(dev(allf), dev(g["mid_ei_sorted"])).cpu().numpy()
    np.testing.assert_allclose(got, g["mid_ew_sorted"], rtol=0, atol=TOL)       # the reference's own weights


def test_median_and_threshold_edges(mmf):
    g = load_golden("g2_threshold.npz")
    for N in (2, 8, 64):
        X, P = g[f"N{N}_X"], g[f"N{N}_P"]
        K = mmf.sim_dense_combined(dev(X), dev(P), 1.0, 1.0)
        np.testing.assert_allclose(K.cpu().numpy(), oracle.sim_dense_combined(X, P, 1.0, 1.0), rtol=0, atol=TOL)
        med = float(mmf.offdiag_lower_median(K))
        assert med == oracle.offdiag_lower_median(K.cpu().numpy())
        for ratio in (0.0, 0.5, 1.0, 2.0):
            ei, ew = mmf.threshold_edges(K, med * ratio)
            assert np.array_equal(ei.cpu().numpy(), g[f"N{N}_r{ratio}_ei"])      # the reference's own edges
            np.testing.assert_allclose(ew.cpu().numpy(), g[f"N{N}_r{ratio}_ew"], rtol=0, atol=TOL)
    K = torch.randn(700, 700, device="cuda")
    assert float(mmf.offdiag_lower_median(K)) == oracle.offdiag_lower_median(K.cpu().numpy())
    ei, ew = mmf.threshold_edges(K, 1.0)
    rei, rew = oracle.threshold_edges(K.cpu().numpy(), 1.0)
    assert np.array_equal(ei.cpu().numpy(), rei) and np.array_equal(ew.cpu().numpy(), rew)
