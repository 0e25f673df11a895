"""Property tests of the HIP path: run-to-run determinism, k-prefix, invariance to the scan precision,
the column split count and the preparation path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mmf():
    import multimodal_fusion_amd as m
    assert torch.cuda.is_available()
    return m


def make(n, d, seed, unit=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, d), generator=g, device="cuda")
    return x / x.norm(dim=1, keepdim=True) if unit else x * 2.0


def test_determinism_and_precision_invariance(mmf):
    X = make(20000, 384, 1)
    ref = mmf.simtopk(X, metric="cosine", k=6, precision="exact")
    for prec in ("fast", "fast_bf16", "fast", "auto"):
        for splits in (0, 1, 4):
            out = mmf.simtopk(X, metric="cosine", k=6, precision=prec, col_splits=splits)
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), (prec, splits)


@pytest.mark.parametrize("metric", ["dot", "neg_sq_l2", "rbf"])
def test_k_prefix_and_unnormalised_metrics(mmf, metric):
    X = make(6000, 200, 2, unit=False) * (0.05 if metric == "rbf" else 1.0)
    prev = None
    for k in (1, 2, 5, 7, 10, 11, 15, 16, 19):
        idx, val = mmf.simtopk(X, metric=metric, lam=0.5, k=k)
        if prev is not None:
            assert torch.equal(idx[:, :prev[0].shape[1]], prev[0]) and torch.equal(val[:, :prev[1].shape[1]], prev[1])
        prev = (idx, val)
    ex = mmf.simtopk(X, metric=metric, lam=0.5, k=19, precision="exact")   # k + self <= 11: 15-entry lists; 12..20: 16-entry lists
    assert torch.equal(ex[0], prev[0])
    if metric != "rbf":
        assert torch.equal(ex[1], prev[1])
    else:
        assert torch.allclose(ex[1], prev[1], rtol=0, atol=1e-5)


def test_rectangular_symmetry_of_scores(mmf):
    # score(x_i, y_j) reported from X's side and from Y's side is the same float (chains commute)
    X, Y = make(3000, 96, 3, unit=False), make(2500, 96, 4, unit=False)
    ix, vx = mmf.simtopk(X, Y, metric="neg_sq_l2", k=1)
    iy, vy = mmf.simtopk(Y, X, metric="neg_sq_l2", k=1)
    j = ix[:, 0]
    mutual = iy[j, 0] == torch.arange(X.shape[0], device="cuda")      # mutual nearest neighbours
    assert int(mutual.sum()) > 50
    assert torch.equal(vx[mutual, 0], vy[j[mutual], 0])


def test_dense_equals_topk_of_dense(mmf):
    X = make(1500, 64, 5, unit=False)
    K = mmf.sim_dense(X, metric="cosine")
    Kd = K.clone()
    Kd.fill_diagonal_(-float("inf"))
    v, i = torch.topk(Kd, 4, dim=1)
    idx, val = mmf.simtopk(X, metric="cosine", k=4)
    assert torch.equal(val, v)                                        # same canonical floats out of both kernels
    ties = (v[:, :-1] == v[:, 1:]).any(dim=1)
    assert torch.equal(idx[~ties], i[~ties])


@pytest.mark.parametrize("flag_rows", [1, 5, 48, 49, 300])
@pytest.mark.parametrize("metric,dtype", [("cosine", torch.float32), ("neg_sq_l2", torch.float32), ("rbf", torch.float16)])
def test_flagged_rows_take_the_exact_paths(mmf, monkeypatch, flag_rows, metric, dtype):
    """Rows the fast path cannot certify are redone exactly: up to 48 of them by the row-vs-all kernels,
    more by the matrix-core rescan.  MMF_DEBUG_FLAG_ROWS flags the first rows artificially."""
    X = (make(9000, 120, 7, unit=False) * (0.05 if metric == "rbf" else 1.0)).to(dtype)
    ref = mmf.simtopk(X, metric=metric, lam=0.5, k=4, precision="exact")
    monkeypatch.setenv("MMF_DEBUG_FLAG_ROWS", str(flag_rows))
    idx, val, st = mmf.simtopk(X, metric=metric, lam=0.5, k=4, precision="fast", return_stats=True)
    assert st["fallback_rows"] >= flag_rows
    assert torch.equal(idx, ref[0]) and torch.equal(val, ref[1])
    Y = (make(5000, 120, 8, unit=False) * (0.05 if metric == "rbf" else 1.0)).to(dtype)       # rectangular, offsets
    r2 = mmf.simtopk(X[:2000], Y, metric=metric, lam=0.5, k=4, precision="exact", exclude_self=True, row_offset=1000, col_offset=900)
    o2 = mmf.simtopk(X[:2000], Y, metric=metric, lam=0.5, k=4, precision="fast", exclude_self=True, row_offset=1000, col_offset=900)
    assert torch.equal(o2[0], r2[0]) and torch.equal(o2[1], r2[1])


def test_random_configurations_against_the_oracle(mmf):
    """Seeded sweep over shapes the parametrised tests do not enumerate: ragged n / m / d, every metric and dtype,
    offsets, self-exclusion on rectangular overlaps, k up to the list limits, forced column splits."""
    import oracle
    rng = np.random.RandomState(20261004)
    metrics = ["cosine", "dot", "neg_sq_l2", "rbf"]
    dtypes = [torch.float32, torch.float16, torch.bfloat16]
    for case in range(36):
        n = int(rng.randint(1, 700))
        m = int(rng.randint(40, 3000))
        d = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 100, 129, 255, 256, 300, 512, 513, 777, 1024, 1100]))
        k = int(rng.randint(1, 23))
        metric = metrics[case % 4]
        dt = dtypes[(case // 4) % 3]
        excl = bool(rng.randint(0, 2))
        ro, co = int(rng.randint(0, 50)), int(rng.randint(0, 50))
        splits = int(rng.choice([0, 0, 1, 2, 8]))
        k = min(k, m - 1)
        scale = 0.05 if metric == "rbf" else 1.0
        g = torch.Generator(device="cuda").manual_seed(1000 + case)
        X = (torch.randn((n, d), generator=g, device="cuda") * scale).to(dt)
        Y = (torch.randn((m, d), generator=g, device="cuda") * scale).to(dt)
        if case % 5 == 0:
            Y[: min(n, m)] = X[: min(n, m)]                                     # duplicates: exact ties, self columns
        idx, val = mmf.simtopk(X, Y, metric=metric, lam=0.7, k=k, exclude_self=excl, row_offset=ro, col_offset=co,
                               col_splits=splits)
        ri, rv = oracle.simtopk(X.float().cpu().numpy(), Y.float().cpu().numpy(), metric=metric, lam=0.7, k=k,
                                exclude_self=excl, row_offset=ro, col_offset=co)
        tag = (case, n, m, d, k, metric, dt, excl, ro, co, splits)
        assert np.array_equal(idx.cpu().numpy(), ri), tag
        if metric == "rbf":
            assert np.allclose(val.cpu().numpy(), rv, rtol=0, atol=1e-5), tag
        else:
            assert np.array_equal(val.cpu().numpy(), rv), tag
