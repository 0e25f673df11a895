"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mmf_hg.h declares, rejects a CPU device id, and the Python mirrors keep the reference's
signatures and fail loudly (no silent CPU fallback) when no GPU is present."""
import inspect
import json
import os
import re
from importlib import import_module

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mmf_hg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmf_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import multimodal_fusion_amd as mmf
    L = mmf._lib.lib()
    names = declared_functions()
    assert {"mmf_simtopk", "mmf_simtopk_ex", "mmf_topk_merge", "mmf_edge_cosine", "mmf_sim_dense",
            "mmf_sim_dense_combined", "mmf_offdiag_lower_median", "mmf_threshold_edges", "mmf_version",
            "mmf_last_error", "mmf_release_workspaces"} <= set(names)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mmf_hg.h but not exported by libmmf_hg.so"
    assert L.mmf_version() == 3


def test_abi_version_is_the_same_everywhere():
    """include/mmf_hg.h, the built library, the ctypes binding and the build check of __graft_entry__ agree (a stale number in
    any of them fails the driver's build step or the first load)."""
    import re
    import multimodal_fusion_amd as mmf
    hdr = open(os.path.join(ROOT, "include", "mmf_hg.h")).read()
    declared = int(re.search(r"#define\s+MMF_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert declared == mmf._lib.ABI_VERSION == mmf._lib.lib().mmf_version()
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "mmf_version() == mmf._lib.ABI_VERSION" in entry       # no literal version number to go stale there


def test_abi_has_no_cpu_path():
    import ctypes
    import multimodal_fusion_amd as mmf
    L = mmf._lib.lib()
    buf = (ctypes.c_float * 16)()
    out_i = (ctypes.c_int64 * 4)()
    out_v = (ctypes.c_float * 4)()
    rc = L.mmf_simtopk(ctypes.cast(buf, ctypes.c_void_p), 4, None, 4, 4, 0, 1, 1.0, 1, 1, 0, 0,
                       ctypes.cast(out_i, ctypes.c_void_p), ctypes.cast(out_v, ctypes.c_void_p), -1, None)
    assert rc == mmf._lib.MMF_E_UNSUPPORTED
    assert b"no CPU path" in L.mmf_last_error()
    rc = L.mmf_sim_dense(ctypes.cast(buf, ctypes.c_void_p), 4, None, 4, 4, 0, 3, 1.0, ctypes.cast(buf, ctypes.c_void_p), -1, None)
    assert rc == mmf._lib.MMF_E_UNSUPPORTED


def test_mirror_signatures_match_the_reference():
    sigs = json.load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))
    for pkg, fns in sigs.items():
        if pkg.endswith("__all__"):
            continue
        mod = import_module("multimodal_fusion_amd." + pkg)
        for name, params in fns.items():
            fn = getattr(mod, name, None) or getattr(import_module("multimodal_fusion_amd." + pkg + ".similarity_kernel"), name)
            ours = [[p.name, None if p.default is inspect._empty else repr(p.default)]
                    for p in inspect.signature(fn).parameters.values()]
            assert ours == params, f"{pkg}.{name}: {ours} != reference {params}"


def test_reference_exports_present():
    b = import_module("multimodal_fusion_amd.build_hypergraph")
    sigs = json.load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))
    assert list(b.__all__) == sigs["build_hypergraph.__all__"] and len(b.__all__) == 17      # build_hypergraph/__init__.py:28-46
    for n in b.__all__:
        assert callable(getattr(b, n)), n
    h = import_module("multimodal_fusion_amd.hypergraph.build_hypergraph")
    for n in ("compute_morphological_similarity", "compute_spatial_similarity", "compute_combined_similarity",
              "build_weighted_hypergraph", "mean_pool_with_similarity"):
        assert callable(getattr(b, n)) and callable(getattr(h, n))
    assert list(h.__all__) == ["compute_morphological_similarity", "compute_spatial_similarity",
                               "compute_combined_similarity", "build_weighted_hypergraph", "mean_pool_with_similarity"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    import multimodal_fusion_amd as mmf
    b = import_module("multimodal_fusion_amd.build_hypergraph")
    x, p = torch.randn(8, 4), torch.rand(8, 2)
    with pytest.raises(RuntimeError, match="no CPU path|needs a ROCm GPU"):
        mmf.simtopk(x, k=2)
    with pytest.raises(RuntimeError, match="needs a ROCm GPU"):
        b.compute_morphological_similarity(x)
    with pytest.raises(RuntimeError, match="needs a ROCm GPU"):
        b.build_weighted_hypergraph(x, p, 1.0, 1.0, 0.5)
    with pytest.raises(RuntimeError, match="needs a ROCm GPU"):
        b.build_hypergraph_knn_kmeans(x, x, None, 2, 2)
    # a5 is plain torch.mean in the reference too
    assert b.mean_pool_with_similarity(x).shape == (1, 4)


def test_f32_ceil_keeps_every_threshold_decision():
    import numpy as np
    c = import_module("multimodal_fusion_amd.build_hypergraph._common")
    rng = np.random.RandomState(0)
    for _ in range(2000):
        thr = float(rng.rand() * rng.choice([1e-3, 1.0, 50.0]))
        t32 = np.float32(c.f32_ceil(thr))
        assert float(t32) >= thr
        below = np.nextafter(t32, np.float32(-np.inf))
        assert float(below) < thr                      # t32 is the smallest float32 >= thr
        for K in (below, t32, np.nextafter(t32, np.float32(np.inf))):
            assert (float(K) < thr) == (K < t32)       # the reference's double compare == our f32 compare
