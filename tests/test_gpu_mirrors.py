"""The reference-signature mirrors on the GPU against the arrays the reference itself produced
(tests/golden/*.npz).  These read like the tests the reference never had: call the function with
the reference's arguments, compare with the reference's outputs."""
import json
from importlib import import_module

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def bh():
    import multimodal_fusion_amd  # noqa: F401
    assert torch.cuda.is_available()
    return import_module("multimodal_fusion_amd.build_hypergraph")


@pytest.fixture(scope="module")
def bh2():
    import multimodal_fusion_amd  # noqa: F401
    return import_module("multimodal_fusion_amd.hypergraph.build_hypergraph")


def T(a, cuda=False):
    t = torch.from_numpy(np.asarray(a))
    return t.cuda() if cuda else t


@pytest.mark.parametrize("N,D", [(2, 4), (64, 32), (256, 128)])
@pytest.mark.parametrize("on_gpu", [False, True])
def test_dense_similarities(bh, N, D, on_gpu):
    g = load_golden("g1_dense.npz")
    X, P2, P3 = (T(g[f"N{N}_D{D}_{n}"], on_gpu) for n in ("X", "P2", "P3"))
    for lam in ((0.5, 1.0, 2.0) if N <= 64 else (1.0,)):
        Kh = bh.compute_morphological_similarity(X, lam)
        assert Kh.device == X.device and Kh.dtype == torch.float32 and Kh.shape == (N, N)
        np.testing.assert_allclose(Kh.cpu().numpy(), g[f"N{N}_D{D}_lam{lam}_Kh"], rtol=0, atol=TOL)
        np.testing.assert_allclose(bh.compute_spatial_similarity(P2, lam).cpu().numpy(), g[f"N{N}_D{D}_lam{lam}_Kg2"], rtol=0, atol=TOL)
        np.testing.assert_allclose(bh.compute_spatial_similarity(P3, lam).cpu().numpy(), g[f"N{N}_D{D}_lam{lam}_Kg3"], rtol=0, atol=TOL)
        np.testing.assert_allclose(bh.compute_combined_similarity(X, P2, lam, lam).cpu().numpy(), g[f"N{N}_D{D}_lam{lam}_K2"], rtol=0, atol=TOL)
        np.testing.assert_allclose(bh.compute_combined_similarity(X, P3, lam, 2.0 * lam).cpu().numpy(), g[f"N{N}_D{D}_lam{lam}_K3"], rtol=0, atol=TOL)
    Z = torch.zeros(N, 2, device=X.device)       # dummy positions, preprocess_hypergraph.py:59
    np.testing.assert_allclose(bh.compute_combined_similarity(X, Z).cpu().numpy(), g[f"N{N}_D{D}_Kzero"], rtol=0, atol=TOL)


@pytest.mark.parametrize("N", [2, 8, 64])
def test_build_weighted_hypergraph_and_data(bh, bh2, N):
    g = load_golden("g2_threshold.npz")
    X, P = T(g[f"N{N}_X"]), T(g[f"N{N}_P"])
    for ratio in (0.0, 0.5, 1.0, 2.0):
        ei, ew = bh.build_weighted_hypergraph(X, P, 1.0, 1.0, ratio)
        assert ei.dtype == torch.int64 and ei.is_contiguous() and ew.dtype == torch.float32 and ei.device.type == "cpu"
        assert np.array_equal(ei.numpy(), g[f"N{N}_r{ratio}_ei"])
        np.testing.assert_allclose(ew.numpy(), g[f"N{N}_r{ratio}_ew"], rtol=0, atol=TOL)
        ei2, ew2 = bh.build_weighted_hypergraph(X.cuda(), P.cuda(), 1.0, 1.0, ratio)
        assert ei2.is_cuda and np.array_equal(ei2.cpu().numpy(), g[f"N{N}_r{ratio}_ei"])
    d1 = bh.build_hypergraph_data(X, P, 1.0, 1.0, 0.5, True)
    d2 = bh2.build_hypergraph_data(X, P, 1.0, 1.0, 0.5, True)
    assert sorted(d1) == ["edge_attr", "edge_index", "pooled_feature", "pos", "x"]
    assert sorted(d2) == ["edge_attr", "edge_index", "pooled_features", "pos", "x"]     # the second copy's key
    assert np.array_equal(d1["edge_index"].numpy(), g[f"N{N}_data_ei"]) and np.array_equal(d2["edge_index"].numpy(), g[f"N{N}_data_ei"])
    np.testing.assert_allclose(d1["edge_attr"].numpy(), g[f"N{N}_data_ew"], rtol=0, atol=TOL)
    np.testing.assert_allclose(d1["pooled_feature"].numpy(), g[f"N{N}_data_pool"], atol=1e-7)
    assert "pooled_feature" not in bh.build_hypergraph_data(X, P, 1.0, 1.0, 0.5, False)


def test_build_weighted_hypergraph_errors(bh):
    g = load_golden("g2_threshold.npz")
    assert json.loads(str(g["errors_json"])) == {"ratio_none": "TypeError", "n1": "ValueError"}
    with pytest.raises(TypeError):          # the reference's default ratio=None blows up at median * None
        bh.build_weighted_hypergraph(torch.randn(4, 8), torch.rand(4, 2))
    with pytest.raises(ValueError, match="greater than 1"):
        bh.build_weighted_hypergraph(torch.randn(1, 8), torch.rand(1, 2), 1.0, 1.0, 0.5)
    ei, ew = bh.build_weighted_hypergraph(torch.randn(6, 8), torch.rand(6, 2), 1.0, 1.0, float("inf"))   # nothing survives
    assert ei.shape == (2, 0) and ew.shape == (0,) and ei.dtype == torch.int64


@pytest.mark.parametrize("N,M", [(8, 12), (100, 37)])
def test_compute_wsi_tma_similarity(bh, N, M):
    g = load_golden("g3_cross.npz")
    A, B = T(g[f"N{N}_M{M}_A"]), T(g[f"N{N}_M{M}_B"])
    for lam in ("1", "0.5"):
        S, st = bh.compute_wsi_tma_similarity(A, torch.rand(N, 2), B, float(lam), 3.0)
        assert S.shape == (N, M) and S.device.type == "cpu"
        np.testing.assert_allclose(S.numpy(), g[f"N{N}_M{M}_S_lam{lam}"], rtol=0, atol=TOL)
        np.testing.assert_allclose([st[k] for k in ("mean", "std", "min", "max", "median")], g[f"N{N}_M{M}_stats_lam{lam}"], rtol=0, atol=1e-5)
        json.dumps(st)


@pytest.mark.parametrize("tag", ["small", "mid", "zero"])
def test_build_hypergraph_knn_kmeans(bh, tag, kmeans_backend):
    g = load_golden("g5_knn_kmeans.npz")
    W, Tm = T(g[f"{tag}_W"]), T(g[f"{tag}_T"])
    k, H = int(g[f"{tag}_k"]), int(g[f"{tag}_H"])
    ei, ew, st = bh.build_hypergraph_knn_kmeans(W, Tm, np.zeros(W.shape[0], dtype=np.int64), k, H)
    ei_ref, ew_ref = g[f"{tag}_ei_sorted"], g[f"{tag}_ew_sorted"]
    assert ei.dtype == torch.int64 and ei.is_contiguous() and bool((ei[0] < ei[1]).all())
    json.dumps(st)
    if tag == "zero":
        # the zero row is at distance ||x_j|| = 1 from EVERY other row: which k of them sklearn reports is
        # unspecified; all other rows and all weights must match
        e = ei.numpy()
        keep, keep_ref = ~((e[0] == 3) | (e[1] == 3)), ~((ei_ref[0] == 3) | (ei_ref[1] == 3))
        assert np.array_equal(e[:, keep], ei_ref[:, keep_ref])
        assert np.all(ew.numpy()[~keep] == 0.0)
    else:
        assert np.array_equal(ei.numpy(), ei_ref)
        np.testing.assert_allclose(ew.numpy(), ew_ref, rtol=0, atol=TOL)
        assert st["num_edges"] == int(g[f"{tag}_num_edges"]) and st["k"] == k and st["num_hyperedges"] == H
    with pytest.raises(ValueError, match="n_neighbors"):
        bh.build_hypergraph_knn_kmeans(W[:3], Tm[:2], None, 5, 2)


def test_aggregate_and_group(bh, kmeans_backend):
    g = load_golden("g1_dense.npz")
    X, P = T(g["N256_D128_X"]), T(g["N256_D128_P2"])
    sf, sp, st, K = bh.aggregate_wsi_super_patches(X, P, 8)
    assert sf.shape == (8, 128) and sp.shape == (8, 2) and K.shape == (256, 256)
    np.testing.assert_allclose(K.numpy(), g["N256_D128_lam1.0_K2"], rtol=0, atol=TOL)
    from sklearn.cluster import KMeans
    labels = KMeans(n_clusters=8, random_state=42, n_init=10).fit_predict(X.numpy())
    for c in range(8):
        np.testing.assert_allclose(sf[c].numpy(), X[labels == c].mean(0).numpy(), atol=1e-6)
    intra = [K[np.ix_(labels == c, labels == c)][~np.eye(int((labels == c).sum()), dtype=bool)].mean().item()
             for c in range(8) if (labels == c).sum() > 1]
    assert abs(st["avg_intra_cluster_similarity"] - float(np.mean(intra))) < 1e-5
    json.dumps(st)
    S, _ = bh.compute_wsi_tma_similarity(sf, sp, X[:40])
    lab, gst = bh.group_by_similarity(S, 3)
    assert lab.shape == (8,) and sum(gst["group_sizes"]) == 8
    json.dumps(gst)
    with pytest.raises(ValueError):
        bh.group_by_similarity(S, 3, method="spectral")


def test_mean_pool_variants(bh, bh2):
    g = load_golden("g7_pool.npz")
    np.testing.assert_allclose(bh.mean_pool_with_similarity(T(g["X"])).numpy(), g["pool1"], atol=1e-7)
    np.testing.assert_allclose(bh2.mean_pool_with_similarity(T(g["X"]), T(g["P"]), 1.0, 1.0).numpy(), g["pool2"], atol=1e-7)


def test_streaming_threshold_builder_equals_materialised(monkeypatch):
    """SURVEY f2 at scale: median + threshold edges from K recomputed in row panels == the materialised path."""
    import multimodal_fusion_amd as mmf
    from multimodal_fusion_amd.build_hypergraph import similarity_kernel as sk
    ops = mmf.ops
    g = torch.Generator().manual_seed(5)
    for N, D, pr in ((3000, 64, 700), (1025, 33, 128), (600, 16, 600)):
        F = (torch.randn((N, D), generator=g) * 0.1).cuda()
        P = (torch.rand((N, 2), generator=g) * 3).cuda()
        K = ops.sim_dense_combined(F, P, 0.7, 0.2)
        med = ops.offdiag_lower_median(K)
        med_s = ops.combined_offdiag_median(F, P, 0.7, 0.2, pr)
        assert torch.equal(med, med_s), (N, D, pr)
        for ratio in (0.5, 1.0, 1.5):
            thr = sk.f32_ceil(float(med) * ratio)
            ei, ew = ops.threshold_edges(K, thr)
            ei_s, ew_s = ops.combined_threshold_edges(F, P, thr, 0.7, 0.2, pr)
            assert torch.equal(ei, ei_s) and torch.equal(ew, ew_s), (N, D, pr, ratio)
    # the mirror switches over by size alone
    F = (torch.randn((900, 32), generator=g) * 0.1)
    P = torch.rand((900, 2), generator=g)
    ref = sk.build_weighted_hypergraph(F, P, 1.0, 1.0, 1.0)
    monkeypatch.setattr(sk, "STREAM_BYTES", 1024)
    monkeypatch.setattr(sk, "PANEL_ROWS", 256)
    out = sk.build_weighted_hypergraph(F, P, 1.0, 1.0, 1.0)
    assert torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1])
    with pytest.raises(ValueError):
        sk.build_weighted_hypergraph(F[:1], P[:1], 1.0, 1.0, 1.0)
