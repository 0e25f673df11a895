"""Property tests (hypothesis) of the oracle — the self-consistency rows of SURVEY.md §4: symmetry of the
self-similarity, the k-prefix property, rank equivalence of the metrics on unit-norm rows, invariance to
column panels and row shards, padding-independence."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle


def data(seed, n, d, unit=True):
    x = np.random.RandomState(seed).randn(n, d).astype(np.float32)
    if unit:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(3, 90), d=st.integers(1, 70))
def test_dense_self_similarity_is_symmetric_bitwise(seed, n, d):
    X = data(seed, n, d, unit=False)
    for metric in ("dot", "cosine", "neg_sq_l2", "rbf"):
        K = oracle.sim_dense(X, metric=metric, lam=0.3)
        assert np.array_equal(K, K.T), metric           # fmaf chains commute in (a, b); (n_i + n_j) commutes


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(8, 120), d=st.integers(2, 64), k=st.integers(1, 6))
def test_k_prefix_property(seed, n, d, k):
    X = data(seed, n, d)
    i1, v1 = oracle.simtopk(X, metric="cosine", k=k)
    i2, v2 = oracle.simtopk(X, metric="cosine", k=k + 1)
    assert np.array_equal(i1, i2[:, :k]) and np.array_equal(v1, v2[:, :k])


@settings(max_examples=20, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(10, 150), d=st.integers(8, 96))
def test_metrics_rank_equivalent_on_unit_rows(seed, n, d):
    X = data(seed, n, d)
    ref, _ = oracle.simtopk(X, metric="cosine", k=3)
    D = (X.astype(np.float64) @ X.astype(np.float64).T)
    np.fill_diagonal(D, -np.inf)
    srt = -np.sort(-D, axis=1)
    clear = (srt[:, :3] - srt[:, 1:4]).min(axis=1) > 1e-5          # rows without a near-tie in the top 4
    for metric in ("dot", "neg_sq_l2", "rbf"):
        idx, _ = oracle.simtopk(X, metric=metric, lam=1.0, k=3)
        assert np.array_equal(idx[clear], ref[clear]), metric


@settings(max_examples=20, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(6, 100), m=st.integers(7, 130), d=st.integers(1, 40),
       cut=st.floats(0.1, 0.9), k=st.integers(1, 5))
def test_column_panels_and_row_shards(seed, n, m, d, cut, k):
    X, Y = data(seed, n, d, unit=False), data(seed + 1, m, d, unit=False)
    k = min(k, max(1, int(m * min(cut, 1 - cut))))
    fi, fv = oracle.simtopk(X, Y, metric="neg_sq_l2", k=k)
    c = max(k, min(m - k, int(m * cut)))
    a = oracle.simtopk(X, Y[:c], metric="neg_sq_l2", k=k)
    b = oracle.simtopk(X, Y[c:], metric="neg_sq_l2", k=k, col_offset=c)
    mi, mv = oracle.topk_merge(a[0], a[1], b[0], b[1])
    assert np.array_equal(mi, fi) and np.array_equal(mv, fv)
    r = n // 2
    top = oracle.simtopk(X[:r], Y, metric="neg_sq_l2", k=k)
    bot = oracle.simtopk(X[r:], Y, metric="neg_sq_l2", k=k, row_offset=r)
    assert np.array_equal(np.concatenate([top[0], bot[0]]), fi)


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(4, 60), d=st.integers(1, 30), pad=st.integers(1, 40))
def test_zero_padding_of_the_feature_dim_changes_nothing(seed, n, d, pad):
    X = data(seed, n, d, unit=False)
    Xp = np.concatenate([X, np.zeros((n, pad), np.float32)], axis=1)
    for metric in ("dot", "cosine", "neg_sq_l2"):
        assert np.array_equal(oracle.sim_dense(X, metric=metric), oracle.sim_dense(Xp, metric=metric))
