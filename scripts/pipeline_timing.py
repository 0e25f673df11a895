"""Wall time of the reference-shaped pipeline steps at several sizes, both KMeans backends."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import multimodal_fusion_amd  # noqa
b = import_module("multimodal_fusion_amd.build_hypergraph")
pp = import_module("multimodal_fusion_amd.build_hypergraph.preprocess_hypergraph")
km = import_module("multimodal_fusion_amd.kmeans")

def t(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, out

rng = np.random.RandomState(0)
for N, D, S in ((4096, 128, 100), (16384, 512, 100), (65536, 512, 100)):
    cent = rng.randn(200, D).astype(np.float32)
    W = torch.from_numpy((cent[rng.randint(0, 200, N)] * 0.3 + 0.05 * rng.randn(N, D)).astype(np.float32)).cuda()
    P = torch.rand(N, 2).cuda()
    T = torch.from_numpy((cent[rng.randint(0, 200, 256)] * 0.3 + 0.05 * rng.randn(256, D)).astype(np.float32)).cuda()
    for backend in ("device", "sklearn"):
        pp.set_kmeans_backend(backend)
        b.aggregate_wsi_super_patches(W[:512], P[:512], 8)      # warm-up: kernels loaded, workspaces grown
        _sf, _sp, _st, _K = b.aggregate_wsi_super_patches(W[:2048], P[:2048], S, 0.5, 1.0)
        _sim, _ = b.compute_wsi_tma_similarity(_sf, _sp, T, 0.5, 1.0)
        _gl, _ = b.group_by_similarity(_sim, 10)
        b.build_hypergraph_knn_kmeans(_sf, T, _gl, 5, 10)
        ms_k, _ = t(lambda: pp._kmeans_labels(W, S))
        if N <= 16384:
            ms_a, (sf, sp, st, K) = t(lambda: b.aggregate_wsi_super_patches(W, P, S, 0.5, 1.0))
        else:
            ms_a, sf = float("nan"), km.kmeans_fit_predict(W, S)[1] if backend == "device" else None
            if sf is None:
                continue
            sp = torch.rand(S, 2).cuda()
        ms_s, (sim, sst) = t(lambda: b.compute_wsi_tma_similarity(sf, sp, T, 0.5, 1.0))
        ms_g, (gl, gst) = t(lambda: b.group_by_similarity(sim, 10))
        ms_h, (ei, ew, hst) = t(lambda: b.build_hypergraph_knn_kmeans(sf, T, gl, 5, 10))
        print(f"N={N} D={D} S={S} {backend:8s}: kmeans({S}) {ms_k:8.1f} ms | aggregate {ms_a:8.1f} | wsi_tma {ms_s:6.1f} | group {ms_g:6.1f} | knn_kmeans {ms_h:6.1f} ms  edges {ei.shape[1]}", flush=True)
