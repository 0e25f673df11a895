"""Whole-result parity where the numbers are quoted.

The 16-bit scan is exact only through its margin proof: every column that can be in the canonical
top-k must survive the approximate filter.  Here the result of the fast path (f16 and bf16 operands) is compared with
the exact f32 scan over EVERY row — indices and scores, torch.equal — at the BASELINE configurations' full sizes,
and both with the CPU oracle on rows spread over the whole matrix.  The exact scan forms canonical keys straight out
of v_mfma_f32_32x32x2_f32 and is itself oracle-checked bit for bit in test_gpu_parity.py / test_gpu_properties.py."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mmf():
    import multimodal_fusion_amd as m
    assert torch.cuda.is_available()
    return m


def make(n, d, seed, block=65536):
    out = torch.empty((n, d), dtype=torch.float32, device="cuda")
    for b in range(0, n, block):
        g = torch.Generator(device="cuda").manual_seed(seed + b)
        x = torch.randn((min(block, n - b), d), generator=g, device="cuda", dtype=torch.float32)
        out[b:b + x.shape[0]] = x / x.norm(dim=1, keepdim=True)
    return out


def oracle_on_rows(X, Y, rows, k, metric, exclude_self, lam=1.0):
    """Oracle top-k of the given (scattered) rows in ONE call: the rows are packed into a small matrix, ranked
    against all columns WITHOUT self exclusion at k + 1, and self is then dropped by identity — the ranking is a
    total order (key desc, id asc), so this equals the oracle's own exclude_self result."""
    Yh = (X if Y is None else Y).float().cpu().numpy()
    Xs = X[rows].float().cpu().numpy()
    kk = k + (1 if exclude_self else 0)
    ri, rv = oracle.simtopk(Xs, Yh, metric=metric, lam=lam, k=kk, exclude_self=False)
    if not exclude_self:
        return ri, rv
    oi = np.empty((len(rows), k), dtype=np.int64)
    ov = np.empty((len(rows), k), dtype=np.float32)
    rr = rows.cpu().numpy()
    for a in range(len(rr)):
        keep = ri[a] != rr[a]
        oi[a] = ri[a][keep][:k]
        ov[a] = rv[a][keep][:k]
    return oi, ov


def spread_rows(n, count=256):
    """`count` rows spread over all row blocks of the matrix, at varying positions inside the 256-query tiles."""
    step = n // count
    i = torch.arange(count, device="cuda")
    return (i * step + (i * 37) % step).long()


def whole_result(mmf, X, Y, metric, k, exclude_self, precisions=("fast", "fast_bf16"), lam=1.0, oracle_rows=256):
    ei, ev, est = mmf.simtopk(X, Y, metric=metric, lam=lam, k=k, exclude_self=exclude_self, precision="exact", return_stats=True)
    assert est["precision_used"] == 1
    for prec in precisions:
        fi, fv, st = mmf.simtopk(X, Y, metric=metric, lam=lam, k=k, exclude_self=exclude_self, precision=prec, return_stats=True)
        assert st["precision_used"] == (2 if prec == "fast" else 3)
        bad = (fi != ei).any(dim=1) | (fv != ev).any(dim=1)
        assert not bool(bad.any()), f"{prec}: {int(bad.sum())} of {X.shape[0]} rows differ from the exact scan, first {int(torch.nonzero(bad)[0])}"
    rows = spread_rows(X.shape[0], oracle_rows)
    oi, ov = oracle_on_rows(X, Y, rows, k, metric, exclude_self, lam)
    assert np.array_equal(ei[rows].cpu().numpy(), oi), "indices differ from the oracle"
    if metric == "rbf":
        np.testing.assert_allclose(ev[rows].cpu().numpy(), ov, rtol=0, atol=1e-5)    # expf: device vs libm, ulps
    else:
        assert np.array_equal(ev[rows].cpu().numpy(), ov), "scores differ from the oracle"
    return ei, ev


def test_c4_headline_workload_every_row(mmf):
    """BASELINE C4 / the bench workload: N = 262144, d = 512, cosine, k = 5, self excluded."""
    from test_gpu_configs import check_against_torch, check_properties
    X = make(262144, 512, 1234)
    ei, ev = whole_result(mmf, X, None, "cosine", 5, True)
    check_properties(ei, ev, 262144, 262144, 0, True)
    check_against_torch(X, None, ei, ev, 5, True)


@pytest.mark.parametrize("metric", ["cosine", "neg_sq_l2", "rbf"])
def test_c2_every_row(mmf, metric):
    X = make(65536, 512, 99)
    whole_result(mmf, X, None, metric, 5, True, oracle_rows=128)


def test_c3_cross_modal_every_row(mmf):
    X, Y = make(65536, 512, 1234), make(65536, 512, 4321)
    whole_result(mmf, X, Y, "cosine", 5, False, oracle_rows=128)


def test_d1024_fp16_features_every_row(mmf):
    """The d <= 1024 variant of the scan (split-k wave pairs; BASELINE C5's shape) at N = 131072, fp16 features."""
    X = make(131072, 1024, 5).half()
    whole_result(mmf, X, None, "cosine", 5, True, oracle_rows=64)


def test_k16_and_clustered_rows_every_row(mmf):
    """The 16-entry-list variant (k + self = 17) and near-duplicate data (overflow lists) against the exact scan."""
    X = make(65536, 512, 77)
    whole_result(mmf, X, None, "cosine", 16, True, precisions=("fast",), oracle_rows=64)
    g = torch.Generator(device="cuda").manual_seed(7)
    centers = make(512, 512, 3)
    assign = torch.randint(0, 512, (65536,), generator=g, device="cuda")
    Xc = centers[assign] + 0.01 * torch.randn((65536, 512), generator=g, device="cuda") / 512 ** 0.5
    Xc = Xc / Xc.norm(dim=1, keepdim=True)
    whole_result(mmf, Xc, None, "neg_sq_l2", 5, True, oracle_rows=64)


def test_c5_one_rank_of_eight_every_row(mmf):
    """BASELINE C5 (N = 1048576, d = 1024, fp16 features) as rank 5 of 8 computes it: 131072 local rows against all
    1 M columns — the split-k 16-bit scan against the exact f32 scan over every local row (2.7e14 flop each way)."""
    N, d, P, r = 1048576, 1024, 8, 5
    X = torch.empty((N, d), dtype=torch.float16, device="cuda")
    for b in range(0, N, 65536):
        g = torch.Generator(device="cuda").manual_seed(5000 + b)
        blk = torch.randn((65536, d), generator=g, device="cuda", dtype=torch.float32)
        X[b:b + 65536] = (blk / blk.norm(dim=1, keepdim=True)).half()
    lo, hi = r * (N // P), (r + 1) * (N // P)
    fi, fv, st = mmf.simtopk(X[lo:hi], X, metric="cosine", k=5, exclude_self=True, row_offset=lo, precision="fast", return_stats=True)
    assert st["precision_used"] == 2
    ei, ev = mmf.simtopk(X[lo:hi], X, metric="cosine", k=5, exclude_self=True, row_offset=lo, precision="exact")
    bad = (fi != ei).any(dim=1) | (fv != ev).any(dim=1)
    assert not bool(bad.any()), f"{int(bad.sum())} of {hi - lo} rows differ from the exact scan"


def test_k32_wide_lists_every_row(mmf):
    """The 32-entry-list variant of the 16-bit scan (k + self = 33) against the exact scan, Gaussian and clustered."""
    X = make(65536, 512, 78)
    whole_result(mmf, X, None, "cosine", 32, True, oracle_rows=32)
    g = torch.Generator(device="cuda").manual_seed(8)
    centers = make(2048, 256, 4)
    assign = torch.randint(0, 2048, (32768,), generator=g, device="cuda")
    Xc = centers[assign] + 0.03 * torch.randn((32768, 256), generator=g, device="cuda") / 256 ** 0.5
    Xc = Xc / Xc.norm(dim=1, keepdim=True)
    whole_result(mmf, Xc, None, "neg_sq_l2", 24, True, precisions=("fast",), oracle_rows=32)


def test_column_range_above_4gib_splits_by_itself(mmf):
    """N = 2.4 M columns at d = 1024 are 4.9 GiB of 16-bit operands: more than the tile DMA's 32-bit offsets address from
    one base, so the library raises the column splits on its own (round 2: an error asking the caller for col_splits).
    512 query rows against all columns: the 16-bit scan against the exact scan, every row."""
    M, d, n = 2400000, 1024, 512
    Y = torch.empty((M, d), dtype=torch.float16, device="cuda")
    for b in range(0, M, 100000):
        g = torch.Generator(device="cuda").manual_seed(9000 + b)
        blk = torch.randn((min(100000, M - b), d), generator=g, device="cuda", dtype=torch.float32)
        Y[b:b + blk.shape[0]] = (blk / blk.norm(dim=1, keepdim=True)).half()
    X = Y[1200000:1200000 + n]
    fi, fv, st = mmf.simtopk(X, Y, metric="cosine", k=5, exclude_self=True, row_offset=1200000, precision="fast", return_stats=True)
    assert st["precision_used"] == 2 and st["col_splits"] >= 2 and st["fallback_rows"] == 0, st
    ei, ev = mmf.simtopk(X, Y, metric="cosine", k=5, exclude_self=True, row_offset=1200000, precision="exact")
    assert torch.equal(fi, ei) and torch.equal(fv, ev)


def test_d1024_k16_on_the_split_k_kernel_every_row(mmf):
    """k + self = 17 at d = 1024: round 2 ran this on a one-wave-per-SIMD kernel that spilled registers; it now takes the split-k
    pair kernel with 15-entry lists (threshold in the lower half of the pair's merged lists).  Every row against the exact scan."""
    X = make(32768, 1024, 11).half()
    whole_result(mmf, X, None, "cosine", 16, True, precisions=("fast",), oracle_rows=32)
    whole_result(mmf, make(16384, 700, 12), None, "neg_sq_l2", 19, True, precisions=("fast",), oracle_rows=16)


def test_clustered_bench_workload_every_row(mmf):
    """`bench.py --data clustered` (N = 262144 rows in 2048 tight clusters: every row's margin band is its whole cluster, 129
    candidates per row): the grouped matrix-core re-rank of the rows with overflow lists (rerank_group_kernel) against the
    exact scan over every row, f16 and bf16 operands, and 64 oracle rows."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_rows
    X = make_rows(0, 262144, 512, torch.device("cuda", 0), data="clustered")
    whole_result(mmf, X, None, "cosine", 5, True, oracle_rows=64)
    _, _, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", return_stats=True)
    assert st["fallback_rows"] == 0 and st["candidates"] > 100 * 262144, st
