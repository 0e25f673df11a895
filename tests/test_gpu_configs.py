"""BASELINE.json configs as parity cases.

C1 N=4096  d=128  self cosine k-NN                      full oracle comparison
C2 N=65536 d=512  self cosine + top-k, 1 GPU            full size: oracle on sampled row blocks (bitwise) +
C3 65536 x 65536  two-modality cross similarity           independent torch check on 2048 rows + order/self properties
C4 N=262144 d=512 row-sharded                           shard(P=8, rank 3) == rows of the unsharded result, bit for bit
C5 d=1024 fp16 features                                  small N against the oracle (AUTO routes d > 512 to the exact scan)
"""
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


def make(n, d, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)


def check_properties(idx, val, n_rows, m, row_offset, exclude_self):
    assert idx.shape == val.shape and idx.dtype == torch.int64 and val.dtype == torch.float32
    assert int(idx.min()) >= 0 and int(idx.max()) < m
    if exclude_self:
        rows = torch.arange(n_rows, device=idx.device)[:, None] + row_offset
        assert not bool((idx == rows).any())
    dv = val[:, 1:] - val[:, :-1]
    assert bool((dv <= 0).all())                                   # scores descending
    ties = dv == 0
    assert bool((idx[:, 1:][ties] > idx[:, :-1][ties]).all())      # equal scores: ascending column id
    s, _ = torch.sort(idx, dim=1)
    assert bool((s[:, 1:] != s[:, :-1]).all())                     # no column twice


def check_against_oracle_blocks(X, Y, idx, val, k, exclude_self, metric="cosine", blocks=3, rows=16):
    Yh = (X if Y is None else Y).cpu().numpy()
    n = X.shape[0]
    for b in range(blocks):
        lo = 0 if b == 0 else (n - rows if b == blocks - 1 else (n // blocks) * b + 5)
        ri, rv = oracle.simtopk(X[lo:lo + rows].cpu().numpy(), Yh, metric=metric, k=k, exclude_self=exclude_self, row_offset=lo)
        assert np.array_equal(idx[lo:lo + rows].cpu().numpy(), ri), f"rows {lo}..: indices differ from the oracle"
        assert np.array_equal(val[lo:lo + rows].cpu().numpy(), rv), f"rows {lo}..: scores differ from the oracle"


def check_against_torch(X, Y, idx, val, k, exclude_self, sample=2048):
    Yt = X if Y is None else Y
    n = X.shape[0]
    rows = torch.linspace(0, n - 1, sample, device=X.device).long().unique()
    S = X[rows] @ Yt.t()                                             # independent path: rocBLAS f32 GEMM
    if exclude_self:
        S[torch.arange(rows.numel(), device=X.device), rows] = -float("inf")
    tv, ti = torch.topk(S, k + 1, dim=1)
    np.testing.assert_allclose(val[rows].cpu().numpy(), tv[:, :k].cpu().numpy(), rtol=0, atol=1e-5)
    clear = (tv[:, :-1] - tv[:, 1:]).min(dim=1).values > 2e-5        # rows whose top-(k+1) has no near-tie
    assert float(clear.float().mean()) > 0.9
    assert bool((idx[rows][clear] == ti[:, :k][clear]).all())


def test_c1_quick_rebuild_shape(mmf):
    X = make(4096, 128, 1234)
    ridx, rval = oracle.simtopk(X.cpu().numpy(), metric="cosine", k=5)
    for prec in ("fast", "exact"):
        idx, val = mmf.simtopk(X, metric="cosine", k=5, precision=prec)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval)
    # the same neighbours through the Euclidean k-NN the reference asks sklearn for (unit-norm rows)
    idx2, _ = mmf.simtopk(X, metric="neg_sq_l2", k=5)
    assert np.array_equal(idx2.cpu().numpy(), ridx)


@pytest.mark.parametrize("metric", ["cosine", "rbf"])
def test_c2_single_modality_full_size(mmf, metric):
    X = make(65536, 512, 1234)
    idx, val, st = mmf.simtopk(X, metric=metric, lam=1.0, k=5, return_stats=True)
    assert st["precision_used"] == 2
    check_properties(idx, val, 65536, 65536, 0, True)
    if metric == "cosine":
        check_against_oracle_blocks(X, None, idx, val, 5, True)
        check_against_torch(X, None, idx, val, 5, True)
    else:
        ri, rv = oracle.simtopk(X[:16].cpu().numpy(), X.cpu().numpy(), metric="rbf", lam=1.0, k=5, exclude_self=True)
        assert np.array_equal(idx[:16].cpu().numpy(), ri)
        np.testing.assert_allclose(val[:16].cpu().numpy(), rv, rtol=0, atol=1e-5)
        ci, _ = mmf.simtopk(X, metric="cosine", k=5)              # rank-equivalent on unit-norm rows
        assert float((ci == idx).all(dim=1).float().mean()) > 0.999


def test_c3_two_modality_cross_full_size(mmf):
    X, Y = make(65536, 512, 1234), make(65536, 512, 4321)
    idx, val = mmf.simtopk(X, Y, metric="cosine", k=5)
    check_properties(idx, val, 65536, 65536, 0, False)
    check_against_oracle_blocks(X, Y, idx, val, 5, False)
    check_against_torch(X, Y, idx, val, 5, False)
    # the dense WSI x TMA matrix of the reference (direct-difference RBF) agrees on a corner block
    S = mmf.sim_dense(X[:256], Y[:512], metric="rbf_direct", lam=1.0).cpu().numpy()
    np.testing.assert_allclose(S, oracle.sim_dense(X[:256].cpu().numpy(), Y[:512].cpu().numpy(), metric="rbf_direct", lam=1.0), rtol=0, atol=1e-5)


def test_c4_row_sharded_equals_unsharded(mmf):
    N = 262144
    X = make(N, 512, 99)
    full_i, full_v = mmf.simtopk(X, metric="cosine", k=5)
    check_properties(full_i, full_v, N, N, 0, True)
    check_against_oracle_blocks(X, None, full_i, full_v, 5, True, blocks=2, rows=8)
    from importlib import import_module
    dmod = import_module("multimodal_fusion_amd.distributed")
    for P, r in ((8, 3), (2, 1)):
        lo, hi = dmod.shard_bounds(N, P, r)
        i, v, st = mmf.simtopk(X[lo:hi], X, metric="cosine", k=5, exclude_self=True, row_offset=lo, return_stats=True)
        assert torch.equal(i, full_i[lo:hi]) and torch.equal(v, full_v[lo:hi]), (P, r)
        # a shard that is NOT a view of the gathered matrix takes the other preparation path: same bits
        i2, v2 = mmf.simtopk(X[lo:hi].clone(), X, metric="cosine", k=5, exclude_self=True, row_offset=lo)
        assert torch.equal(i2, i) and torch.equal(v2, v)


def test_c5_fp16_features_d1024(mmf):
    X = make(3000, 1024, 7).half()
    ridx, rval = oracle.simtopk(X.float().cpu().numpy(), metric="cosine", k=5)
    for prec, used in (("auto", 2), ("exact", 1)):    # d = 1024 runs on the 4-wave variant of the 16-bit scan
        idx, val, st = mmf.simtopk(X, metric="cosine", k=5, precision=prec, return_stats=True)
        assert st["precision_used"] == used
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(val.cpu().numpy(), rval), prec
    Xb = make(40000, 1024, 9).half()                                           # bigger, through properties + oracle blocks
    idx, val, st = mmf.simtopk(Xb, metric="cosine", k=5, return_stats=True)
    assert st["precision_used"] == 2
    check_properties(idx, val, 40000, 40000, 0, True)
    check_against_oracle_blocks(Xb.float(), None, idx, val, 5, True, blocks=2, rows=8)
    Xc = make(5000, 700, 10)                                                    # ragged d between the padded sizes
    ri, rv = oracle.simtopk(Xc.cpu().numpy(), metric="neg_sq_l2", k=4)
    idx, val, st = mmf.simtopk(Xc, metric="neg_sq_l2", k=4, return_stats=True)
    assert st["precision_used"] == 2 and np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(val.cpu().numpy(), rv)
    Xh = make(20000, 512, 8).half()                                            # fp16 features on the 16-bit scan
    idx, val, st = mmf.simtopk(Xh, metric="cosine", k=5, return_stats=True)
    assert st["precision_used"] == 2
    check_properties(idx, val, 20000, 20000, 0, True)
    check_against_oracle_blocks(Xh.float(), None, idx, val, 5, True)


@pytest.mark.parametrize("metric", ["cosine", "neg_sq_l2", "dot"])
def test_phase_api_equals_single_call(mmf, metric):
    """mmf_row_scalars + mmf_prep_rows per shard, concatenated as an all-gather would, then
    mmf_simtopk_prepared for one shard == the rows of the single-call result, bit for bit."""
    N, d, P, k = 8192, 256, 4, 5
    X = make(N, d, 31) * (1.0 if metric == "cosine" else 3.0)
    full_i, full_v = mmf.simtopk(X, metric=metric, k=k)
    ops = mmf.ops
    dp = ops.padded_dim(d)
    rows = N // P
    maxn = torch.zeros(1, device="cuda")
    scal = [torch.empty(rows, device="cuda") for _ in range(P)]
    for r in range(P):
        ops.row_scalars(X[r * rows:(r + 1) * rows], metric, scal[r], maxn)      # shared maximum == all-reduce MAX
    m_pad = (N + 255) // 256 * 256
    Z = torch.zeros((m_pad + 256, dp), dtype=torch.float16, device="cuda")
    side = {n_: torch.full((m_pad + 256,), float("-inf") if n_ == "cb" else 0.0, device="cuda") for n_ in ("scal", "zn", "rn", "un", "cb")}
    max4 = torch.zeros(4, device="cuda")
    for r in range(P):
        sl = slice(r * rows, (r + 1) * rows)
        z = torch.empty((rows, dp), dtype=torch.float16, device="cuda")
        zn, rn, un, cb = (torch.empty(rows, device="cuda") for _ in range(4))
        ops.prep_rows(X[sl], metric, "f16", scal[r], maxn, z, zn, rn, un, cb, max4)
        Z[sl] = z
        side["scal"][sl], side["zn"][sl], side["rn"][sl], side["un"][sl], side["cb"][sl] = scal[r], zn, rn, un, cb
    for r in (0, 2, 3):
        lo, hi = r * rows, (r + 1) * rows
        q = dict(Z=Z[lo:], **{n_: side[n_][lo:] for n_ in side})
        ev = torch.cuda.Event()
        ev.record()
        for order in ("off", "on"):                       # the scan's query order (csrc/mmf_order.hip) changes nothing
            i, v, st = ops.simtopk_prepared(X[lo:hi], X, q, dict(Z=Z, **side), m_pad, max4, metric=metric, k=k,
                                            exclude_self=True, row_offset=lo, wait_event=ev, return_stats=True, query_order=order)
            assert st["query_order"] == (order == "on")
            assert torch.equal(i, full_i[lo:hi]) and torch.equal(v, full_v[lo:hi]), (metric, r, order)


@pytest.mark.parametrize("metric,S,splits", [("cosine", 4, 0), ("neg_sq_l2", 2, 4), ("dot", 1, 8), ("rbf", 4, 1)])
def test_paneled_scan_equals_single_call(mmf, metric, S, splits):
    """mmf_simtopk_panels on the chunk layout of a pipelined all-gather (chunk c = rows [c*rows/S, (c+1)*rows/S)
    of every rank, rank-major) == the rows of the single-call result, bit for bit."""
    N, d, P, k = 16384, 200, 4, 6
    X = make(N, d, 41) * (1.0 if metric == "cosine" else (0.05 if metric == "rbf" else 3.0))
    X[300:360] = X[300]                       # 60 exact copies: their columns overflow the lane lists into the rows'
    X[5000:5060] = X[300]                     # overflow lists, whose ids go through the panel -> global column map
    X[7000:7050] = X[7000]                    # 50 copies: they FIT the many lists of a paneled scan (no overflow entry), but each of these
                                              # rows carries 49 candidates after pruning: the first re-rank pass hands them to the second
    full_i, full_v = mmf.simtopk(X, metric=metric, lam=0.5, k=k)
    ex_i, _ = mmf.simtopk(X, metric=metric, lam=0.5, k=k, precision="exact")
    assert torch.equal(full_i, ex_i)
    ops = mmf.ops
    dp = ops.padded_dim(d)
    rows = N // P
    seg = rows // S
    maxn = torch.zeros(1, device="cuda")
    scal = torch.empty(N, device="cuda")
    for r in range(P):
        ops.row_scalars(X[r * rows:(r + 1) * rows], metric, scal[r * rows:(r + 1) * rows], maxn)
    max4 = torch.zeros(4, device="cuda")
    Z = torch.zeros((N + 256, dp), dtype=torch.float16, device="cuda")
    zn, rn, un, cb = (torch.zeros(N + 256, device="cuda") for _ in range(4))
    for r in range(P):
        sl = slice(r * rows, (r + 1) * rows)
        ops.prep_rows(X[sl], metric, "f16", scal[sl], maxn, Z[sl], zn[sl], rn[sl], un[sl], cb[sl], max4)
    m_c = P * seg
    m_pad = (m_c + 255) // 256 * 256
    panels = []
    for c in range(S):
        Zc = torch.zeros((m_pad + 256, dp), dtype=torch.float16, device="cuda")
        cbc = torch.full((m_pad + 256,), float("-inf"), device="cuda")
        Zc[:m_c] = Z[:N].view(P, S, seg, dp)[:, c].reshape(m_c, dp)
        cbc[:m_c] = cb[:N].view(P, S, seg)[:, c].reshape(m_c)
        ev = torch.cuda.Event()
        ev.record()
        panels.append(dict(Z=Zc, cb=cbc, m=m_c, m_pad=m_pad, seg_len=seg, seg_stride=rows, id_base=c * seg, event=ev))
    for r in (0, 3):
        lo, hi = r * rows, (r + 1) * rows
        q = dict(Z=Z[lo:], scal=scal[lo:], zn=zn[lo:], rn=rn[lo:], un=un[lo:], cb=cb[lo:])
        for order in ("off", "on"):
            i, v, st = ops.simtopk_panels(X[lo:hi], X, q, scal, panels, max4, metric=metric, lam=0.5, k=k, exclude_self=True,
                                          row_offset=lo, col_splits=splits, return_stats=True, query_order=order)
            assert st["query_order"] == (order == "on")
            assert torch.equal(i, full_i[lo:hi]), (metric, r, order)
            if metric == "rbf":
                assert torch.allclose(v, full_v[lo:hi], rtol=0, atol=1e-5)
            else:
                assert torch.equal(v, full_v[lo:hi]), (metric, r, order)
    with pytest.raises(ValueError):
        bad = [dict(panels[0], m=m_c - seg)] + panels[1:]                      # panels no longer cover Y
        ops.simtopk_panels(X[:rows], X, q, scal, bad, max4, metric=metric, lam=0.5, k=k)


def test_c5_full_size_shard_of_eight(mmf):
    """BASELINE config 5 at full size: N = 1048576, d = 1024, fp16 features.  One GPU runs what rank 5 of 8
    would (131072 local rows against all 1M columns, 2.8e14 flop) and the oracle checks sampled rows."""
    N, d, P, r = 1048576, 1024, 8, 5
    X = torch.empty((N, d), dtype=torch.float16, device="cuda")
    for b in range(0, N, 65536):
        g = torch.Generator(device="cuda").manual_seed(5000 + b)
        blk = torch.randn((65536, d), generator=g, device="cuda", dtype=torch.float32)
        X[b:b + 65536] = (blk / blk.norm(dim=1, keepdim=True)).half()
    lo, hi = r * (N // P), (r + 1) * (N // P)
    idx, val, st = mmf.simtopk(X[lo:hi], X, metric="cosine", k=5, exclude_self=True, row_offset=lo, return_stats=True)
    assert st["precision_used"] == 2 and st["fallback_rows"] < 64
    check_properties(idx, val, hi - lo, N, lo, True)
    Xh = X.float().cpu().numpy()
    for off in (0, 70000, hi - lo - 8):
        ri, rv = oracle.simtopk(Xh[lo + off:lo + off + 8], Xh, metric="cosine", k=5, exclude_self=True, row_offset=lo + off)
        assert np.array_equal(idx[off:off + 8].cpu().numpy(), ri) and np.array_equal(val[off:off + 8].cpu().numpy(), rv)
