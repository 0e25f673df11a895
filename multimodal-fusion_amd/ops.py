"""Torch-tensor front end of the C ABI: device pointers, current stream, output allocation.

Every function requires tensors on a ROCm device (``tensor.is_cuda``); nothing here computes on
the host.  The reference-signature mirrors in ``build_hypergraph`` sit on top of this module.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def _need_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; this op runs only on a ROCm device "
                           "(the package has no CPU path)")


def _feat(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dim() != 2:
        raise ValueError(f"{what}: expected a 2-D [N, D] tensor, got shape {tuple(t.shape)}")
    if t.dtype not in _DT:
        t = t.float()
    return t.contiguous()


def _stream(dev: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _metric(m) -> int:
    if isinstance(m, str):
        if m not in _lib.METRICS:
            raise ValueError(f"unknown metric {m!r}")
        return _lib.METRICS[m]
    return int(m)


def simtopk(X: torch.Tensor, Y: Optional[torch.Tensor] = None, *, metric="cosine", lam: float = 1.0, k: int = 5,
            exclude_self: Optional[bool] = None, row_offset: int = 0, col_offset: int = 0, precision: str = "auto",
            profile: bool = False, col_splits: int = 0, return_stats: bool = False, query_order: str = "auto"):
    """Fused similarity + per-row top-k (mmf_simtopk_ex).  Returns (idx int64 [n,k], val f32 [n,k]).
    query_order ("auto" / "off" / "on"): whether the 16-bit scan takes near-duplicate rows next to each other
    (include/mmf_hg.h MMF_QUERY_ORDER_*; the result does not depend on it)."""
    X = _feat(X, "simtopk X")
    _need_gpu(X, "simtopk")
    if Y is not None:
        Y = _feat(Y, "simtopk Y")
        if Y.device != X.device or Y.dtype != X.dtype or Y.shape[1] != X.shape[1]:
            raise ValueError("simtopk: X and Y must share device, dtype and feature dim")
    if exclude_self is None:
        exclude_self = Y is None
    n, d = X.shape
    m = n if Y is None else Y.shape[0]
    idx = torch.empty((n, k), dtype=torch.int64, device=X.device)
    val = torch.empty((n, k), dtype=torch.float32, device=X.device)
    opts = _lib.SimtopkOpts(_lib.PRECISIONS[precision], int(profile), int(col_splits), _lib.QUERY_ORDERS[query_order], None)
    stats = _lib.SimtopkStats()
    rc = _lib.lib().mmf_simtopk_ex(_p(X), n, _p(Y), m, d, _DT[X.dtype], _metric(metric), float(lam), int(k),
                                   int(bool(exclude_self)), int(row_offset), int(col_offset), _p(idx), _p(val),
                                   ctypes.byref(opts), ctypes.byref(stats), X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_simtopk")
    if return_stats:
        return idx, val, stats.as_dict()
    return idx, val


def last_query_order(n: int) -> torch.Tensor:
    """Diagnostics: scan position -> row of the most recent simtopk call that reordered its n queries (mmf_debug_query_order)."""
    out = torch.empty((n,), dtype=torch.int32)
    _lib.check(_lib.lib().mmf_debug_query_order(ctypes.c_void_p(out.data_ptr()), ctypes.c_int64(n)), "mmf_debug_query_order")
    return out


def sim_dense(X: torch.Tensor, Y: Optional[torch.Tensor] = None, *, metric="rbf", lam: float = 1.0) -> torch.Tensor:
    X = _feat(X, "sim_dense X")
    _need_gpu(X, "sim_dense")
    if Y is not None:
        Y = _feat(Y, "sim_dense Y").to(X.dtype)
        if Y.device != X.device or Y.shape[1] != X.shape[1]:
            raise ValueError("sim_dense: X and Y must share device and feature dim")
    n, d = X.shape
    m = n if Y is None else Y.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=X.device)
    rc = _lib.lib().mmf_sim_dense(_p(X), n, _p(Y), m, d, _DT[X.dtype], _metric(metric), float(lam), _p(out),
                                  X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_sim_dense")
    return out


def _stats_dict(out: torch.Tensor) -> dict:
    h = out.to(torch.float32).cpu().tolist()        # the reference's values are `.item()`s of f32 tensors
    return {"mean": h[0], "std": h[1], "min": h[2], "max": h[3], "median": h[4]}


def sim_dense_stats(X: torch.Tensor, Y: Optional[torch.Tensor] = None, *, metric="rbf_direct", lam: float = 1.0,
                    store: bool = True, panel_rows: int = 0):
    """(S [n, m] f32 or None, {'mean','std','min','max','median'}) from mmf_sim_dense_stats.  store=False
    (rbf_direct only) never materialises S."""
    X = _feat(X, "sim_dense_stats X")
    _need_gpu(X, "sim_dense_stats")
    if Y is not None:
        Y = _feat(Y, "sim_dense_stats Y").to(X.dtype)
        if Y.device != X.device or Y.shape[1] != X.shape[1]:
            raise ValueError("sim_dense_stats: X and Y must share device and feature dim")
    n, d = X.shape
    m = n if Y is None else Y.shape[0]
    if n < 1 or m < 1:
        raise ValueError("sim_dense_stats: empty matrix")
    out = torch.empty((n, m), dtype=torch.float32, device=X.device) if store else None
    st = torch.empty((5,), dtype=torch.float64, device=X.device)
    rc = _lib.lib().mmf_sim_dense_stats(_p(X), n, _p(Y), m, d, _DT[X.dtype], _metric(metric), float(lam), _p(out), _p(st),
                                        int(panel_rows), X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_sim_dense_stats")
    return out, _stats_dict(st)


def sim_dense_combined(F: torch.Tensor, P: torch.Tensor, lambda_h: float = 1.0, lambda_g: float = 1.0) -> torch.Tensor:
    F = _feat(F, "sim_dense_combined features").float()
    P = _feat(P, "sim_dense_combined positions").float()
    _need_gpu(F, "sim_dense_combined")
    if P.device != F.device or P.shape[0] != F.shape[0]:
        raise ValueError("sim_dense_combined: features and positions must share device and N")
    n, d = F.shape
    out = torch.empty((n, n), dtype=torch.float32, device=F.device)
    rc = _lib.lib().mmf_sim_dense_combined(_p(F), _p(P), n, d, P.shape[1], float(lambda_h), float(lambda_g), _p(out),
                                           F.device.index or 0, _stream(F.device))
    _lib.check(rc, "mmf_sim_dense_combined")
    return out


def edge_cosine(X: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    X = _feat(X, "edge_cosine X")
    _need_gpu(X, "edge_cosine")
    ei = edge_index.to(device=X.device, dtype=torch.int64).contiguous()
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError("edge_cosine: edge_index must be [2, E]")
    E = ei.shape[1]
    if E and (int(ei.min()) < 0 or int(ei.max()) >= X.shape[0]):
        raise ValueError("edge_cosine: edge_index out of range")
    out = torch.empty((E,), dtype=torch.float32, device=X.device)
    rc = _lib.lib().mmf_edge_cosine(_p(X), X.shape[0], X.shape[1], _DT[X.dtype], _p(ei), E, _p(out),
                                    X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_edge_cosine")
    return out


def topk_merge(ia: torch.Tensor, va: torch.Tensor, ib: torch.Tensor, vb: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_gpu(ia, "topk_merge")
    ia, ib = ia.contiguous().long(), ib.contiguous().long()
    va, vb = va.contiguous().float(), vb.contiguous().float()
    if ia.shape != ib.shape or ia.shape != va.shape or ia.shape != vb.shape or ia.dim() != 2:
        raise ValueError("topk_merge: all four inputs must be [n, k]")
    n, k = ia.shape
    io = torch.empty_like(ia)
    vo = torch.empty_like(va)
    rc = _lib.lib().mmf_topk_merge(_p(ia), _p(va), _p(ib), _p(vb), n, k, _p(io), _p(vo), ia.device.index or 0,
                                   _stream(ia.device))
    _lib.check(rc, "mmf_topk_merge")
    return io, vo


def offdiag_lower_median(K: torch.Tensor) -> torch.Tensor:
    """Lower median of the off-diagonal entries of a square f32 matrix; returns a 0-d device tensor (one sweep over K for
    n >= 2049, include/mmf_hg.h "Lower medians")."""
    _need_gpu(K, "offdiag_lower_median")
    K = K.contiguous().float()
    if K.dim() != 2 or K.shape[0] != K.shape[1]:
        raise ValueError("offdiag_lower_median: K must be square")
    out = torch.empty((), dtype=torch.float32, device=K.device)
    rc = _lib.lib().mmf_offdiag_lower_median(_p(K), K.shape[0], _p(out), K.device.index or 0, _stream(K.device))
    _lib.check(rc, "mmf_offdiag_lower_median")
    return out


def lower_median(v: torch.Tensor) -> torch.Tensor:
    """torch.median of a flat f32 tensor (lower median); returns a 0-d device tensor.  4 M values or more: one sweep
    (sampled bracket verified by exact counts, include/mmf_hg.h "Lower medians"), else a 4-pass radix select."""
    _need_gpu(v, "lower_median")
    v = v.contiguous().float().reshape(-1)
    if v.numel() < 1:
        raise ValueError("lower_median: empty input")
    out = torch.empty((), dtype=torch.float32, device=v.device)
    rc = _lib.lib().mmf_lower_median(_p(v), v.numel(), _p(out), v.device.index or 0, _stream(v.device))
    _lib.check(rc, "mmf_lower_median")
    return out


def array_stats(v: torch.Tensor) -> dict:
    """{'mean','std','min','max','median'} of a dense f32 tensor as Python floats (the reference's `.item()` values:
    f32-rounded), from one reduction pass + a radix select on the device (mmf_array_stats)."""
    _need_gpu(v, "array_stats")
    v = v.contiguous().float().reshape(-1)
    if v.numel() < 1:
        raise ValueError("array_stats: empty input")
    out = torch.empty((5,), dtype=torch.float64, device=v.device)
    rc = _lib.lib().mmf_array_stats(_p(v), v.numel(), _p(out), v.device.index or 0, _stream(v.device))
    _lib.check(rc, "mmf_array_stats")
    return _stats_dict(out)


def threshold_edges(K: torch.Tensor, threshold: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row-major (i, j) with K[i, j] not below `threshold` -> (edge_index [2,E] int64, weights [E] f32)."""
    _need_gpu(K, "threshold_edges")
    K = K.contiguous().float()
    if K.dim() != 2 or K.shape[0] != K.shape[1]:
        raise ValueError("threshold_edges: K must be square")
    n = K.shape[0]
    cnt = torch.zeros((), dtype=torch.int64, device=K.device)
    L = _lib.lib()
    dev, st = K.device.index or 0, _stream(K.device)
    # pass 1: count + row offsets; pass 2: fill from the offsets (K is read twice in all)
    row_off = torch.empty((n + 1,), dtype=torch.int64, device=K.device)
    _lib.check(L.mmf_threshold_edges_count(_p(K), n, float(threshold), _p(row_off), _p(cnt), dev, st), "mmf_threshold_edges_count")
    E = int(cnt.item())
    ei = torch.empty((2, E), dtype=torch.int64, device=K.device)
    ew = torch.empty((E,), dtype=torch.float32, device=K.device)
    if E:
        _lib.check(L.mmf_threshold_edges_fill(_p(K), n, float(threshold), _p(row_off), _p(ei), _p(ew), E, dev, st),
                   "mmf_threshold_edges_fill")
    return ei, ew


def combined_offdiag_median(F: torch.Tensor, P: torch.Tensor, lambda_h: float = 1.0, lambda_g: float = 1.0,
                            panel_rows: int = 0) -> torch.Tensor:
    """Lower median of the off-diagonal entries of K = K_h * K_g without materialising K (recomputed in row panels)."""
    F = _feat(F, "combined_offdiag_median features").float()
    P = _feat(P, "combined_offdiag_median positions").float()
    _need_gpu(F, "combined_offdiag_median")
    if P.device != F.device or P.shape[0] != F.shape[0]:
        raise ValueError("combined_offdiag_median: features and positions must share device and N")
    out = torch.empty((), dtype=torch.float32, device=F.device)
    rc = _lib.lib().mmf_combined_offdiag_median(_p(F), _p(P), F.shape[0], F.shape[1], P.shape[1], float(lambda_h),
                                                float(lambda_g), int(panel_rows), _p(out), F.device.index or 0,
                                                _stream(F.device))
    _lib.check(rc, "mmf_combined_offdiag_median")
    return out


def combined_threshold_edges(F: torch.Tensor, P: torch.Tensor, threshold: float, lambda_h: float = 1.0,
                             lambda_g: float = 1.0, panel_rows: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """threshold_edges of K = K_h * K_g without materialising K: one sweep to count, one to fill."""
    F = _feat(F, "combined_threshold_edges features").float()
    P = _feat(P, "combined_threshold_edges positions").float()
    _need_gpu(F, "combined_threshold_edges")
    if P.device != F.device or P.shape[0] != F.shape[0]:
        raise ValueError("combined_threshold_edges: features and positions must share device and N")
    n, d = F.shape
    cnt = torch.zeros((), dtype=torch.int64, device=F.device)
    L = _lib.lib()
    args = (_p(F), _p(P), n, d, P.shape[1], float(lambda_h), float(lambda_g), float(threshold), int(panel_rows))
    dev, st = F.device.index or 0, _stream(F.device)
    _lib.check(L.mmf_combined_threshold_edges(*args, None, None, 0, _p(cnt), dev, st), "mmf_combined_threshold_edges")
    E = int(cnt.item())
    ei = torch.empty((2, E), dtype=torch.int64, device=F.device)
    ew = torch.empty((E,), dtype=torch.float32, device=F.device)
    if E:
        _lib.check(L.mmf_combined_threshold_edges(*args, _p(ei), _p(ew), E, _p(cnt), dev, st), "mmf_combined_threshold_edges")
    return ei, ew


# ---------------------------------------------------------------------------------------------------
# phase API of the fast path (row-sharded multi-GPU driver, distributed.py)
# ---------------------------------------------------------------------------------------------------
_OPERAND = {"f16": _lib.F16, "bf16": _lib.BF16}


def padded_dim(d: int) -> int:
    """Padded feature dim of the 16-bit operands; 0 when d is not supported by the 16-bit scan."""
    return int(_lib.lib().mmf_padded_dim(int(d)))


def fast_scan_supported(d: int, k: int, exclude_self: bool = True) -> bool:
    """Whether the 16-bit scan (and with it the phase API) handles this feature dim and k."""
    return bool(_lib.lib().mmf_fast_scan_supported(int(d), int(k), int(bool(exclude_self))))


def row_scalars(X: torch.Tensor, metric, scal: torch.Tensor, max_sq_norm: Optional[torch.Tensor] = None) -> None:
    """scal[i] = canonical n_i (clamped norm for cosine); max_sq_norm[0] is raised to max n_i."""
    X = _feat(X, "row_scalars X")
    _need_gpu(X, "row_scalars")
    rc = _lib.lib().mmf_row_scalars(_p(X), X.shape[0], X.shape[1], _DT[X.dtype], _metric(metric), _p(scal),
                                    _p(max_sq_norm), X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_row_scalars")


def prep_rows(X: torch.Tensor, metric, operand: str, scal: torch.Tensor, max_sq_norm: Optional[torch.Tensor],
              Z: torch.Tensor, zn: torch.Tensor, rn: torch.Tensor, un: torch.Tensor, cb: torch.Tensor,
              maxima: torch.Tensor) -> None:
    """16-bit operands + per-row norms of the rows of X into caller-owned buffers (mmf_prep_rows)."""
    X = _feat(X, "prep_rows X")
    _need_gpu(X, "prep_rows")
    n_pad = Z.shape[0]
    for t in (Z, zn, rn, un, cb):
        if not t.is_contiguous():
            raise ValueError("prep_rows: output buffers must be contiguous")
    rc = _lib.lib().mmf_prep_rows(_p(X), X.shape[0], X.shape[1], _DT[X.dtype], _metric(metric), _OPERAND[operand],
                                  _p(scal), _p(max_sq_norm), _p(Z), n_pad, _p(zn), _p(rn), _p(un), _p(cb), _p(maxima),
                                  X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_prep_rows")


def simtopk_prepared(X: torch.Tensor, Y: torch.Tensor, q: dict, c: dict, m_pad: int, maxima: torch.Tensor, *,
                     operand: str = "f16", metric="cosine", lam: float = 1.0, k: int = 5, exclude_self: bool = True,
                     row_offset: int = 0, col_offset: int = 0, profile: bool = False, col_splits: int = 0,
                     wait_event: Optional[torch.cuda.Event] = None, return_stats: bool = False, query_order: str = "auto"):
    """Scan on prepared operands + exact re-rank (mmf_simtopk_prepared).  q / c: dicts with the tensors
    Z, scal, zn, rn, un, cb of the query / candidate side.  wait_event: recorded when the f32 rows of
    X / Y are complete; the stream waits for it only after the scan."""
    _need_gpu(X, "simtopk_prepared")
    n, d = X.shape
    m = Y.shape[0]
    idx = torch.empty((n, k), dtype=torch.int64, device=X.device)
    val = torch.empty((n, k), dtype=torch.float32, device=X.device)

    def side(dd):
        return _lib.PreparedSide(*(ctypes.c_void_p(dd[key].data_ptr()) for key in ("Z", "scal", "zn", "rn", "un", "cb")))
    qs, cs = side(q), side(c)
    ev = ctypes.c_void_p(wait_event.cuda_event) if wait_event is not None else None
    opts = _lib.SimtopkOpts(_lib.PRECISIONS["fast"], int(profile), int(col_splits), _lib.QUERY_ORDERS[query_order], ev)
    stats = _lib.SimtopkStats()
    rc = _lib.lib().mmf_simtopk_prepared(_p(X), n, _p(Y), m, d, _DT[X.dtype], _metric(metric), float(lam), int(k),
                                         int(bool(exclude_self)), int(row_offset), int(col_offset), ctypes.byref(qs),
                                         ctypes.byref(cs), int(m_pad), _p(maxima), _OPERAND[operand], _p(idx), _p(val),
                                         ctypes.byref(opts), ctypes.byref(stats), X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_simtopk_prepared")
    if return_stats:
        return idx, val, stats.as_dict()
    return idx, val


def simtopk_panels(X: torch.Tensor, Y: torch.Tensor, q: dict, c_scal: torch.Tensor, panels: list, maxima: torch.Tensor, *,
                   operand: str = "f16", metric="cosine", lam: float = 1.0, k: int = 5, exclude_self: bool = True,
                   row_offset: int = 0, col_offset: int = 0, profile: bool = False, col_splits: int = 0,
                   wait_event: Optional[torch.cuda.Event] = None, return_stats: bool = False, query_order: str = "auto"):
    """Paneled scan + exact re-rank (mmf_simtopk_panels).  panels: dicts with Z [m_pad+256, dp], cb [m_pad+256], m,
    m_pad, and optionally seg_len / seg_stride / id_base (panel column -> column of Y) and `event` (a
    torch.cuda.Event the scan of that panel waits for).  q: query-side dict as for simtopk_prepared."""
    _need_gpu(X, "simtopk_panels")
    n, d = X.shape
    m = Y.shape[0]
    idx = torch.empty((n, k), dtype=torch.int64, device=X.device)
    val = torch.empty((n, k), dtype=torch.float32, device=X.device)
    qs = _lib.PreparedSide(*(ctypes.c_void_p(q[key].data_ptr()) for key in ("Z", "scal", "zn", "rn", "un", "cb")))
    arr = (_lib.Panel * len(panels))()
    for i, pn in enumerate(panels):
        ev = pn.get("event")
        arr[i] = _lib.Panel(ctypes.c_void_p(pn["Z"].data_ptr()), ctypes.c_void_p(pn["cb"].data_ptr()), int(pn["m"]),
                            int(pn["m_pad"]), int(pn.get("seg_len", 0)), int(pn.get("seg_stride", 0)),
                            int(pn.get("id_base", 0)), ctypes.c_void_p(ev.cuda_event) if ev is not None else None)
    ev = ctypes.c_void_p(wait_event.cuda_event) if wait_event is not None else None
    opts = _lib.SimtopkOpts(_lib.PRECISIONS["fast"], int(profile), int(col_splits), _lib.QUERY_ORDERS[query_order], ev)
    stats = _lib.SimtopkStats()
    rc = _lib.lib().mmf_simtopk_panels(_p(X), n, _p(Y), m, d, _DT[X.dtype], _metric(metric), float(lam), int(k),
                                       int(bool(exclude_self)), int(row_offset), int(col_offset), ctypes.byref(qs),
                                       _p(c_scal), arr, len(panels), _p(maxima), _OPERAND[operand], _p(idx), _p(val),
                                       ctypes.byref(opts), ctypes.byref(stats), X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_simtopk_panels")
    if return_stats:
        return idx, val, stats.as_dict()
    return idx, val


# ---------------------------------------------------------------------------------------------------
# cluster-shaped steps (mmf_segments.hip): members of each label, per-cluster means, cliques, k-NN pair dedup
# ---------------------------------------------------------------------------------------------------
class Segments:
    """Members of every label, grouped: `order[offsets[c]:offsets[c+1]]` are the rows of segment c, ascending."""
    __slots__ = ("counts", "offsets", "order", "n", "n_segments")

    def __init__(self, counts, offsets, order, n, n_segments):
        self.counts, self.offsets, self.order, self.n, self.n_segments = counts, offsets, order, n, n_segments


def segment_sort(labels: torch.Tensor, n_segments: int) -> Segments:
    """Stable counting sort of integer labels in [0, n_segments) on the device (mmf_segment_sort)."""
    _need_gpu(labels, "segment_sort")
    lab = labels.to(torch.int64).contiguous().reshape(-1)
    n, S = lab.numel(), int(n_segments)
    counts = torch.empty((S,), dtype=torch.int64, device=lab.device)
    offsets = torch.empty((S + 1,), dtype=torch.int64, device=lab.device)
    order = torch.empty((n,), dtype=torch.int64, device=lab.device)
    rc = _lib.lib().mmf_segment_sort(_p(lab), n, S, _p(counts), _p(offsets), _p(order), lab.device.index or 0,
                                     _stream(lab.device))
    _lib.check(rc, "mmf_segment_sort")
    return Segments(counts, offsets, order, n, S)


def segment_mean(X: torch.Tensor, seg: Segments) -> torch.Tensor:
    """[n_segments, d] per-segment mean of the rows of X (f32), summed in member order (mmf_segment_mean)."""
    X = _feat(X, "segment_mean X").float()
    _need_gpu(X, "segment_mean")
    if X.shape[0] != seg.n:
        raise ValueError("segment_mean: X has a different number of rows than the labels")
    out = torch.empty((seg.n_segments, X.shape[1]), dtype=torch.float32, device=X.device)
    rc = _lib.lib().mmf_segment_mean(_p(X), X.shape[0], X.shape[1], _p(seg.order), _p(seg.offsets), seg.n_segments,
                                     _p(out), X.device.index or 0, _stream(X.device))
    _lib.check(rc, "mmf_segment_mean")
    return out


def segment_offdiag_mean(K: torch.Tensor, seg: Segments) -> torch.Tensor:
    """[n_segments] f64: mean of K[i, j] over the ordered pairs i != j inside each segment (NaN below two members)."""
    _need_gpu(K, "segment_offdiag_mean")
    K = K.contiguous().float()
    if K.dim() != 2 or K.shape[0] != K.shape[1] or K.shape[0] != seg.n:
        raise ValueError("segment_offdiag_mean: K must be [n, n] with n = number of labels")
    out = torch.empty((seg.n_segments,), dtype=torch.float64, device=K.device)
    rc = _lib.lib().mmf_segment_offdiag_mean(_p(K), K.shape[0], _p(seg.order), _p(seg.offsets), seg.n_segments, _p(out),
                                             K.device.index or 0, _stream(K.device))
    _lib.check(rc, "mmf_segment_offdiag_mean")
    return out


def clique_pairs(seg: Segments) -> Tuple[torch.Tensor, torch.Tensor]:
    """(lo, hi) int64: every pair a < b inside every segment (mmf_clique_pairs: count, then fill)."""
    dev = seg.order.device
    cnt = torch.zeros((), dtype=torch.int64, device=dev)
    L = _lib.lib()
    args = (_p(seg.order), _p(seg.offsets), seg.n, seg.n_segments)
    _lib.check(L.mmf_clique_pairs(*args, None, None, 0, _p(cnt), dev.index or 0, _stream(dev)), "mmf_clique_pairs")
    E = int(cnt.item())
    lo = torch.empty((E,), dtype=torch.int64, device=dev)
    hi = torch.empty((E,), dtype=torch.int64, device=dev)
    if E:
        _lib.check(L.mmf_clique_pairs(*args, _p(lo), _p(hi), E, _p(cnt), dev.index or 0, _stream(dev)), "mmf_clique_pairs")
    return lo, hi


def knn_pairs(nbr: torch.Tensor, labels: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Undirected duplicate-free (lo, hi) pairs of a [n, k] neighbour table; pairs inside one label are left to the
    cliques (mmf_knn_pairs).  Order unspecified."""
    _need_gpu(nbr, "knn_pairs")
    nbr = nbr.to(torch.int64).contiguous()
    n, k = nbr.shape
    lab = None if labels is None else labels.to(device=nbr.device, dtype=torch.int64).contiguous()
    lo = torch.empty((n * k,), dtype=torch.int64, device=nbr.device)
    hi = torch.empty((n * k,), dtype=torch.int64, device=nbr.device)
    cnt = torch.zeros((), dtype=torch.int64, device=nbr.device)
    rc = _lib.lib().mmf_knn_pairs(_p(nbr), n, k, _p(lab), _p(lo), _p(hi), _p(cnt), nbr.device.index or 0, _stream(nbr.device))
    _lib.check(rc, "mmf_knn_pairs")
    E = int(cnt.item())
    return lo[:E], hi[:E]


def kmeans_fit(X: torch.Tensor, n_clusters: int, first_centres, uniforms, *, max_iter: int = 300, tol: float = 1e-4,
               return_seeds: bool = False):
    """scikit-learn's KMeans fit on the device, decision for decision (mmf_kmeans_fit).  X f32 [n, d] on the GPU;
    first_centres: int64 [n_init] and uniforms: float64 [n_init, n_clusters - 1, trials] are HOST (numpy) arrays holding
    scikit-learn's random stream (multimodal-fusion_amd/kmeans.py draws them).  Returns (labels int64 [n],
    centres f32 [n_clusters, d], info dict[, seeds int64 [n_init, n_clusters]])."""
    import numpy as np
    X = _feat(X, "kmeans_fit X").float()
    _need_gpu(X, "kmeans_fit")
    n, d = X.shape
    first = np.ascontiguousarray(first_centres, dtype=np.int64)
    n_init = int(first.shape[0])
    k = int(n_clusters)
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    if k > 1 and (u.ndim != 3 or u.shape[0] != n_init or u.shape[1] != k - 1):
        raise ValueError("kmeans_fit: uniforms must be [n_init, n_clusters - 1, trials]")
    trials = int(u.shape[2]) if k > 1 else 1
    labels = torch.empty(n, dtype=torch.int64, device=X.device)
    centres = torch.empty((k, d), dtype=torch.float32, device=X.device)
    seeds = torch.empty((n_init, k), dtype=torch.int64, device=X.device) if return_seeds else None
    info = np.zeros(7 + 2 * n_init, dtype=np.float64)
    rc = _lib.lib().mmf_kmeans_fit(_p(X), n, d, k, n_init, trials, ctypes.c_void_p(first.ctypes.data),
                                   ctypes.c_void_p(u.ctypes.data) if k > 1 else None, int(max_iter), float(tol), _p(labels),
                                   _p(centres), _p(seeds), ctypes.c_void_p(info.ctypes.data), X.device.index or 0,
                                   _stream(X.device))
    _lib.check(rc, "mmf_kmeans_fit")
    out_info = {"best_init": int(info[0]), "inertia": float(info[1]), "n_iter": int(info[2]), "tol_abs": float(info[3]),
                "ambiguous_draws": int(info[4]), "ambiguous_trials": int(info[5]), "lockstep_iterations": int(info[6]),
                "inertia_per_init": info[7::2].tolist(), "n_iter_per_init": [int(v) for v in info[8::2]]}
    if return_seeds:
        return labels, centres, out_info, seeds
    return labels, centres, out_info
