"""ctypes binding of libmmf_hg.so (include/mmf_hg.h).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MMF_HG_LIBRARY") or os.path.join(_HERE, "libmmf_hg.so")   # override: a library built elsewhere

MMF_OK, MMF_E_INVALID, MMF_E_UNSUPPORTED, MMF_E_HIP, MMF_E_NOMEM, MMF_E_INTERNAL = 0, -1, -2, -3, -4, -5
DOT, COSINE, NEG_SQ_L2, RBF, RBF_DIRECT = 0, 1, 2, 3, 4
METRICS = {"dot": DOT, "cosine": COSINE, "neg_sq_l2": NEG_SQ_L2, "rbf": RBF, "rbf_direct": RBF_DIRECT}
F32, BF16, F16 = 0, 1, 2
PRECISIONS = {"auto": 0, "exact": 1, "fast": 2, "fast_bf16": 3}
QUERY_ORDERS = {"auto": 0, "off": 1, "on": 2}
ABI_VERSION = 3          # MMF_ABI_VERSION of the include/mmf_hg.h this binding was written against


class SimtopkOpts(ctypes.Structure):
    _fields_ = [("precision", ctypes.c_int), ("profile", ctypes.c_int), ("col_splits", ctypes.c_int),
                ("query_order", ctypes.c_int), ("select_wait_event", ctypes.c_void_p)]


class PreparedSide(ctypes.Structure):
    _fields_ = [("Z", ctypes.c_void_p), ("scal", ctypes.c_void_p), ("zn", ctypes.c_void_p), ("rn", ctypes.c_void_p),
                ("un", ctypes.c_void_p), ("cb", ctypes.c_void_p)]


class Panel(ctypes.Structure):
    _fields_ = [("Z", ctypes.c_void_p), ("cb", ctypes.c_void_p), ("m", ctypes.c_int64), ("m_pad", ctypes.c_int64),
                ("seg_len", ctypes.c_int64), ("seg_stride", ctypes.c_int64), ("id_base", ctypes.c_int64),
                ("ready_event", ctypes.c_void_p)]


class SimtopkStats(ctypes.Structure):
    _fields_ = [("scan_ms", ctypes.c_float), ("prep_ms", ctypes.c_float), ("rerank_ms", ctypes.c_float),
                ("fallback_ms", ctypes.c_float), ("candidates", ctypes.c_int64), ("fallback_rows", ctypes.c_int64),
                ("precision_used", ctypes.c_int), ("col_splits", ctypes.c_int), ("scan_grid", ctypes.c_int),
                ("scan_wait_ms", ctypes.c_float), ("overflow_rows", ctypes.c_int64), ("short_rows", ctypes.c_int64),
                ("near_rows", ctypes.c_int64), ("order_ms", ctypes.c_float), ("query_order", ctypes.c_int)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_ if not f.startswith("reserved")}


_lib = None

EXPORTS = ["mmf_version", "mmf_last_error", "mmf_simtopk", "mmf_simtopk_ex", "mmf_row_scalars", "mmf_prep_rows",
           "mmf_simtopk_prepared", "mmf_simtopk_panels", "mmf_padded_dim", "mmf_fast_scan_supported", "mmf_topk_merge", "mmf_edge_cosine",
           "mmf_sim_dense", "mmf_sim_dense_stats", "mmf_sim_dense_combined", "mmf_offdiag_lower_median", "mmf_threshold_edges", "mmf_threshold_edges_count", "mmf_threshold_edges_fill", "mmf_lower_median", "mmf_array_stats",
           "mmf_segment_sort", "mmf_segment_mean", "mmf_segment_offdiag_mean", "mmf_clique_pairs", "mmf_knn_pairs", "mmf_kmeans_fit", "mmf_combined_offdiag_median", "mmf_combined_threshold_edges",
           "mmf_release_workspaces", "mmf_debug_query_order"]


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or python multimodal-fusion_amd/csrc/build.py).  This package has no CPU fallback.")
    L = ctypes.CDLL(SO_PATH)
    vp, i64, ci, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    L.mmf_version.restype = ci
    L.mmf_last_error.restype = ctypes.c_char_p
    L.mmf_simtopk.argtypes = [vp, i64, vp, i64, i64, ci, ci, f32, ci, ci, i64, i64, vp, vp, ci, vp]
    L.mmf_simtopk_ex.argtypes = [vp, i64, vp, i64, i64, ci, ci, f32, ci, ci, i64, i64, vp, vp,
                                 ctypes.POINTER(SimtopkOpts), ctypes.POINTER(SimtopkStats), ci, vp]
    L.mmf_padded_dim.argtypes = [i64]
    L.mmf_fast_scan_supported.argtypes = [i64, ci, ci]
    L.mmf_row_scalars.argtypes = [vp, i64, i64, ci, ci, vp, vp, ci, vp]
    L.mmf_prep_rows.argtypes = [vp, i64, i64, ci, ci, ci, vp, vp, vp, i64, vp, vp, vp, vp, vp, ci, vp]
    L.mmf_simtopk_prepared.argtypes = [vp, i64, vp, i64, i64, ci, ci, f32, ci, ci, i64, i64,
                                       ctypes.POINTER(PreparedSide), ctypes.POINTER(PreparedSide), i64, vp, ci, vp, vp,
                                       ctypes.POINTER(SimtopkOpts), ctypes.POINTER(SimtopkStats), ci, vp]
    L.mmf_simtopk_panels.argtypes = [vp, i64, vp, i64, i64, ci, ci, f32, ci, ci, i64, i64,
                                     ctypes.POINTER(PreparedSide), vp, ctypes.POINTER(Panel), ci, vp, ci, vp, vp,
                                     ctypes.POINTER(SimtopkOpts), ctypes.POINTER(SimtopkStats), ci, vp]
    L.mmf_topk_merge.argtypes = [vp, vp, vp, vp, i64, ci, vp, vp, ci, vp]
    L.mmf_edge_cosine.argtypes = [vp, i64, i64, ci, vp, i64, vp, ci, vp]
    L.mmf_sim_dense.argtypes = [vp, i64, vp, i64, i64, ci, ci, f32, vp, ci, vp]
    L.mmf_sim_dense_stats.argtypes = [vp, i64, vp, i64, i64, ci, ci, f32, vp, vp, i64, ci, vp]
    L.mmf_sim_dense_combined.argtypes = [vp, vp, i64, i64, i64, f32, f32, vp, ci, vp]
    L.mmf_offdiag_lower_median.argtypes = [vp, i64, vp, ci, vp]
    L.mmf_segment_sort.argtypes = [vp, i64, i64, vp, vp, vp, ci, vp]
    L.mmf_segment_mean.argtypes = [vp, i64, i64, vp, vp, i64, vp, ci, vp]
    L.mmf_segment_offdiag_mean.argtypes = [vp, i64, vp, vp, i64, vp, ci, vp]
    L.mmf_clique_pairs.argtypes = [vp, vp, i64, i64, vp, vp, i64, vp, ci, vp]
    L.mmf_knn_pairs.argtypes = [vp, i64, ci, vp, vp, vp, vp, ci, vp]
    L.mmf_kmeans_fit.argtypes = [vp, i64, i64, i64, i64, ci, vp, vp, ci, ctypes.c_double, vp, vp, vp, vp, ci, vp]
    L.mmf_lower_median.argtypes = [vp, i64, vp, ci, vp]
    L.mmf_array_stats.argtypes = [vp, i64, vp, ci, vp]
    L.mmf_threshold_edges.argtypes = [vp, i64, f32, vp, vp, i64, vp, ci, vp]
    L.mmf_threshold_edges_count.argtypes = [vp, i64, f32, vp, vp, ci, vp]
    L.mmf_threshold_edges_fill.argtypes = [vp, i64, f32, vp, vp, vp, i64, ci, vp]
    L.mmf_combined_offdiag_median.argtypes = [vp, vp, i64, i64, i64, f32, f32, i64, vp, ci, vp]
    L.mmf_combined_threshold_edges.argtypes = [vp, vp, i64, i64, i64, f32, f32, f32, i64, vp, vp, i64, vp, ci, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("mmf_last_error", "mmf_padded_dim"):
            fn.restype = ci
    L.mmf_padded_dim.restype = i64
    L.mmf_last_error.restype = ctypes.c_char_p
    if L.mmf_version() != ABI_VERSION:
        raise RuntimeError(f"{SO_PATH}: ABI version {L.mmf_version()}, this package binds version {ABI_VERSION} "
                           "(include/mmf_hg.h MMF_ABI_VERSION): rebuild the library")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc == MMF_OK:
        return
    msg = lib().mmf_last_error().decode("utf-8", "replace")
    if rc == MMF_E_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (code {rc})")
