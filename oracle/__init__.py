"""CPU oracle for the hypergraph-construction hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``multimodal-fusion_amd/``) never does, and fails loudly when its HIP
library is missing instead of falling back to anything here.

Two layers:

* ``oracle.mmf_oracle.c`` (bound below through ctypes) — the canonical-arithmetic restatement
  (k-ordered ``fmaf`` chains, see ``include/mmf_hg.h``) that the HIP path must match bit for bit
  on indices and to 1e-5 on scores.
* ``oracle.ref_restate`` — a line-by-line torch-CPU restatement of the reference functions
  (``build_hypergraph/similarity_kernel.py``, ``build_hypergraph/preprocess_hypergraph.py``),
  used to pin the C layer to the golden vectors captured from the reference itself.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmmf_oracle.so")
_SRC = os.path.join(_HERE, "mmf_oracle.c")

DOT, COSINE, NEG_SQ_L2, RBF, RBF_DIRECT = 0, 1, 2, 3, 4
METRICS = {"dot": DOT, "cosine": COSINE, "neg_sq_l2": NEG_SQ_L2, "rbf": RBF, "rbf_direct": RBF_DIRECT}

CFLAGS = ["-O3", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-std=gnu11"]


def build(force: bool = False) -> str:
    """Compile oracle/mmf_oracle.c -> oracle/libmmf_oracle.so (gcc).  Idempotent."""
    if (not force) and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    cmd = ["gcc", *CFLAGS, _SRC, "-o", _SO + ".tmp", "-lm"]
    subprocess.run(cmd, check=True)
    os.replace(_SO + ".tmp", _SO)
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, f32, ci = ctypes.c_int64, ctypes.c_float, ctypes.c_int
        vp = ctypes.c_void_p
        L.mmf_oracle_simtopk.argtypes = [vp, i64, vp, i64, i64, ci, f32, ci, ci, i64, i64, vp, vp, ci]
        L.mmf_oracle_simtopk.restype = ci
        L.mmf_oracle_sim_dense.argtypes = [vp, i64, vp, i64, i64, ci, f32, vp, ci]
        L.mmf_oracle_sim_dense.restype = ci
        L.mmf_oracle_sim_dense_combined.argtypes = [vp, vp, i64, i64, i64, f32, f32, vp, ci]
        L.mmf_oracle_sim_dense_combined.restype = ci
        L.mmf_oracle_edge_cosine.argtypes = [vp, i64, i64, vp, i64, vp]
        L.mmf_oracle_edge_cosine.restype = ci
        L.mmf_oracle_topk_merge.argtypes = [vp, vp, vp, vp, i64, ci, vp, vp]
        L.mmf_oracle_topk_merge.restype = ci
        L.mmf_oracle_offdiag_lower_median.argtypes = [vp, i64, vp]
        L.mmf_oracle_offdiag_lower_median.restype = ci
        L.mmf_oracle_threshold_edges.argtypes = [vp, i64, f32, vp, vp, i64]
        L.mmf_oracle_threshold_edges.restype = i64
        L.mmf_oracle_num_threads.restype = ci
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    if hasattr(a, "detach"):  # torch tensor (bf16/f16 are upcast exactly)
        a = a.detach().to("cpu").float().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads() -> int:
    return int(lib().mmf_oracle_num_threads())


def simtopk(X, Y=None, *, metric="cosine", lam=1.0, k=5, exclude_self=None, row_offset=0,
            col_offset=0, nthreads=0):
    """Canonical fused similarity + top-k.  Returns (idx int64 [n,k], val f32 [n,k])."""
    X = _f32(X)
    n, d = X.shape
    if Y is None:
        Yp, m = None, n
        if exclude_self is None:
            exclude_self = True
    else:
        Yc = _f32(Y)
        m = Yc.shape[0]
        if Yc.shape[1] != d:
            raise ValueError("feature dims differ")
        Yp = _ptr(Yc)
        if exclude_self is None:
            exclude_self = False
    idx = np.empty((n, k), dtype=np.int64)
    val = np.empty((n, k), dtype=np.float32)
    mt = METRICS[metric] if isinstance(metric, str) else int(metric)
    rc = lib().mmf_oracle_simtopk(_ptr(X), n, Yp, m, d, mt, float(lam), int(k), int(bool(exclude_self)),
                                  int(row_offset), int(col_offset), _ptr(idx), _ptr(val), int(nthreads))
    if rc == -1:
        raise ValueError("oracle.simtopk: invalid argument")
    if rc != 0:
        raise MemoryError("oracle.simtopk: allocation failed")
    return idx, val


def sim_dense(X, Y=None, *, metric="rbf", lam=1.0, nthreads=0) -> np.ndarray:
    X = _f32(X)
    n, d = X.shape
    if Y is None:
        Yp, m = None, n
    else:
        Yc = _f32(Y)
        m = Yc.shape[0]
        Yp = _ptr(Yc)
    out = np.empty((n, m), dtype=np.float32)
    mt = METRICS[metric] if isinstance(metric, str) else int(metric)
    rc = lib().mmf_oracle_sim_dense(_ptr(X), n, Yp, m, d, mt, float(lam), _ptr(out), int(nthreads))
    if rc != 0:
        raise ValueError("oracle.sim_dense: invalid argument")
    return out


def sim_dense_combined(F, P, lambda_h=1.0, lambda_g=1.0, nthreads=0) -> np.ndarray:
    F = _f32(F)
    P = _f32(P)
    n, d = F.shape
    out = np.empty((n, n), dtype=np.float32)
    rc = lib().mmf_oracle_sim_dense_combined(_ptr(F), _ptr(P), n, d, P.shape[1], float(lambda_h),
                                             float(lambda_g), _ptr(out), int(nthreads))
    if rc != 0:
        raise ValueError("oracle.sim_dense_combined: invalid argument")
    return out


def edge_cosine(X, edge_index) -> np.ndarray:
    X = _f32(X)
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    E = ei.shape[1]
    out = np.empty((E,), dtype=np.float32)
    rc = lib().mmf_oracle_edge_cosine(_ptr(X), X.shape[0], X.shape[1], _ptr(ei), E, _ptr(out))
    if rc != 0:
        raise ValueError("oracle.edge_cosine: invalid argument")
    return out


def topk_merge(ia, va, ib, vb):
    ia = np.ascontiguousarray(ia, dtype=np.int64)
    ib = np.ascontiguousarray(ib, dtype=np.int64)
    va = np.ascontiguousarray(va, dtype=np.float32)
    vb = np.ascontiguousarray(vb, dtype=np.float32)
    n, k = ia.shape
    io = np.empty((n, k), dtype=np.int64)
    vo = np.empty((n, k), dtype=np.float32)
    rc = lib().mmf_oracle_topk_merge(_ptr(ia), _ptr(va), _ptr(ib), _ptr(vb), n, k, _ptr(io), _ptr(vo))
    if rc != 0:
        raise ValueError("oracle.topk_merge: invalid argument")
    return io, vo


def offdiag_lower_median(K) -> float:
    K = _f32(K)
    out = np.empty((1,), dtype=np.float32)
    rc = lib().mmf_oracle_offdiag_lower_median(_ptr(K), K.shape[0], _ptr(out))
    if rc != 0:
        raise ValueError("oracle.offdiag_lower_median: need n >= 2")
    return float(out[0])


def threshold_edges(K, threshold: float):
    K = _f32(K)
    n = K.shape[0]
    cap = n * n
    ei = np.empty((2, cap), dtype=np.int64)
    ew = np.empty((cap,), dtype=np.float32)
    cnt = int(lib().mmf_oracle_threshold_edges(_ptr(K), n, float(threshold), _ptr(ei), _ptr(ew), cap))
    return np.ascontiguousarray(ei[:, :cnt]), ew[:cnt].copy()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
