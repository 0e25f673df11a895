"""Query order of the 16-bit scan (csrc/mmf_order.hip): the scan may take its query rows in any order — near-duplicate rows next
to each other make a wave's hits coincide — and NOTHING in the result may depend on it.  Every case runs with the order forced on
(MMF_QUERY_ORDER_ON; AUTO only tries it from 32768 rows) against the CPU oracle (indices bit-exact, scores bitwise, RBF to 1e-5)
and against the same call with the order off (same bits)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def mmf():
    import multimodal_fusion_amd as m
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return m


def dup_rows(n, d, seed, clusters=None, noise=0.02, scale=1.0):
    """Rows in tight clusters, scattered: what the order is for."""
    rng = np.random.RandomState(seed)
    c = rng.randn(clusters or max(3, n // 40), d).astype(np.float32)
    return ((c[rng.randint(0, len(c), n)] + noise * rng.randn(n, d).astype(np.float32)) * scale).astype(np.float32)


def both(mmf, X, Y, **kw):
    on = mmf.simtopk(X, Y, query_order="on", return_stats=True, **kw)
    off = mmf.simtopk(X, Y, query_order="off", return_stats=True, **kw)
    assert on[2]["query_order"] == 1 and off[2]["query_order"] == 0 and off[2]["near_rows"] == -1
    assert torch.equal(on[0], off[0]) and torch.equal(on[1].view(torch.int32), off[1].view(torch.int32)), "the order changed the result"
    return on


@pytest.mark.parametrize("metric", ["dot", "cosine", "neg_sq_l2", "rbf"])
@pytest.mark.parametrize("n,d,k,precision", [(3000, 64, 5, "fast"), (1111, 200, 9, "fast_bf16"), (300, 40, 3, "fast"), (2500, 512, 16, "fast"),
                                             (1500, 96, 30, "fast"), (1300, 1000, 5, "fast")])
def test_self_similarity_against_the_oracle(mmf, metric, n, d, k, precision):
    X = dup_rows(n, d, n + d + k, scale=0.1 if metric == "rbf" else 1.0)
    idx, val, st = both(mmf, torch.from_numpy(X).cuda(), None, metric=metric, lam=0.5, k=k, precision=precision)
    ridx, rval = oracle.simtopk(X, None, metric=metric, lam=0.5, k=k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    if metric == "rbf":
        np.testing.assert_allclose(val.cpu().numpy(), rval, rtol=0, atol=TOL)
    else:
        assert np.array_equal(val.cpu().numpy(), rval)
    assert st["near_rows"] >= 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_rectangular_with_offsets_and_self_exclusion(mmf, dtype):
    Y = torch.from_numpy(dup_rows(5000, 120, 3)).to(dtype)
    X = Y[torch.from_numpy(np.random.RandomState(4).randint(0, 5000, 2000))].clone()       # queries: copies of candidate rows
    kw = dict(metric="neg_sq_l2", k=6, exclude_self=True, row_offset=1000, col_offset=900)
    idx, val, _ = both(mmf, X.cuda(), Y.cuda(), precision="fast", **kw)
    ridx, rval = oracle.simtopk(X.float().numpy(), Y.float().numpy(), **kw)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)


def test_rows_that_are_a_slice_of_the_candidates(mmf):
    """X = Y[lo:hi] in memory (the row-sharded case): the query-side operands are views into the candidate side, the ordered copy
    is a buffer of its own."""
    Y = torch.from_numpy(dup_rows(6000, 72, 8)).cuda()
    lo, hi = 1500, 4100
    idx, val, _ = both(mmf, Y[lo:hi], Y, metric="cosine", k=5, exclude_self=True, row_offset=lo, precision="fast")
    ridx, rval = oracle.simtopk(Y[lo:hi].cpu().numpy(), Y.cpu().numpy(), metric="cosine", k=5, exclude_self=True, row_offset=lo)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)


@pytest.mark.parametrize("flag_rows", [3, 48, 200])
def test_flagged_rows_with_the_order_on(mmf, monkeypatch, flag_rows):
    """The first scan positions are flagged artificially: with the order on those are OTHER rows than without — each must come back
    from the exact paths in its own place."""
    X = torch.from_numpy(dup_rows(7000, 100, 21)).cuda()
    ref = mmf.simtopk(X, metric="cosine", k=4, precision="exact")
    monkeypatch.setenv("MMF_DEBUG_FLAG_ROWS", str(flag_rows))
    idx, val, st = mmf.simtopk(X, metric="cosine", k=4, precision="fast", query_order="on", return_stats=True)
    assert st["fallback_rows"] >= flag_rows and st["query_order"] == 1
    assert torch.equal(idx, ref[0]) and torch.equal(val, ref[1])


def test_auto_decides_from_the_data(mmf):
    """From 32768 rows AUTO measures: scattered near-duplicates -> ordered, Gaussian rows -> left alone; below that it does not try."""
    g = torch.Generator(device="cuda").manual_seed(3)
    Xg = torch.randn((40000, 64), generator=g, device="cuda")
    c = torch.randn((800, 64), generator=g, device="cuda")
    Xd = c[torch.randint(0, 800, (40000,), generator=g, device="cuda")] + 0.02 * torch.randn((40000, 64), generator=g, device="cuda")
    for X, want in ((Xg, 0), (Xd, 1)):
        a = mmf.simtopk(X, metric="cosine", k=5, precision="fast", return_stats=True)
        b = mmf.simtopk(X, metric="cosine", k=5, precision="fast", query_order="off")
        assert a[2]["query_order"] == want and (a[2]["near_rows"] >= 8 * 128 if want else a[2]["near_rows"] == 0)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    small = mmf.simtopk(Xd[:20000], metric="cosine", k=5, precision="fast", return_stats=True)
    assert small[2]["query_order"] == 0 and small[2]["near_rows"] == -1
    with pytest.raises(KeyError):
        mmf.simtopk(Xg[:100], k=3, query_order="sometimes")


def test_the_recorded_permutation(mmf):
    """mmf_debug_query_order: scan position -> row of the last ordered call is a permutation that puts copies of a row into the same scan
    waves; it is forgotten by the next fast-path call."""
    rng = np.random.RandomState(9)
    base = rng.randn(50, 48).astype(np.float32)
    lab = rng.randint(0, 50, 4000)
    X = torch.from_numpy(base[lab] + 1e-4 * rng.randn(4000, 48).astype(np.float32)).cuda()
    mmf.simtopk(X, metric="cosine", k=3, precision="fast", query_order="on")
    perm = mmf.ops.last_query_order(4000).numpy()
    assert np.array_equal(np.sort(perm), np.arange(4000))
    # 50 groups of ~80 copies: a scan wave (32 consecutive positions) holds one or two of them (32 in row order)
    waves = lab[perm][: 4000 // 32 * 32].reshape(-1, 32)
    distinct = np.array([len(np.unique(w)) for w in waves])
    assert distinct.mean() <= 2.0, distinct.mean()
    mmf.simtopk(X, metric="cosine", k=3, precision="fast", query_order="off")
    with pytest.raises(ValueError):
        mmf.ops.last_query_order(4000)
