"""Device KMeans against scikit-learn's KMeans(random_state=42, n_init=10), the call
the reference makes.  Labels cannot match bit for bit (different seeding stream); the objective must."""
from importlib import import_module

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_kmeans_separable_blobs_same_partition():
    import multimodal_fusion_amd  # noqa: F401
    km = import_module("multimodal_fusion_amd.kmeans")
    from sklearn.cluster import KMeans
    from sklearn.metrics import adjusted_rand_score
    rng = np.random.RandomState(0)
    centers = rng.randn(8, 32).astype(np.float32) * 6.0
    X = np.concatenate([c + rng.randn(300, 32).astype(np.float32) for c in centers], 0)
    ref = KMeans(n_clusters=8, random_state=42, n_init=10).fit(X)
    labels, C, inertia = km.kmeans_fit_predict(torch.from_numpy(X).cuda(), 8)
    assert adjusted_rand_score(ref.labels_, labels.cpu().numpy()) == 1.0
    assert abs(inertia - ref.inertia_) <= 1e-4 * ref.inertia_
    assert labels.dtype == torch.int64 and C.shape == (8, 32)


def test_kmeans_unstructured_data_objective():
    import multimodal_fusion_amd  # noqa: F401
    km = import_module("multimodal_fusion_amd.kmeans")
    from sklearn.cluster import KMeans
    X = np.random.RandomState(1).randn(4000, 64).astype(np.float32)
    ref = KMeans(n_clusters=10, random_state=42, n_init=10).fit(X)
    labels, C, inertia = km.kmeans_fit_predict(torch.from_numpy(X).cuda(), 10)
    assert inertia <= 1.02 * ref.inertia_                         # as good a local optimum, within 2 %
    # the reported inertia is the objective of the returned partition
    d2 = ((X - C.cpu().numpy()[labels.cpu().numpy()]) ** 2).sum()
    assert abs(d2 - inertia) <= 1e-3 * inertia
    assert int(torch.bincount(labels, minlength=10).min()) > 0
    with pytest.raises(ValueError):
        km.kmeans_fit_predict(torch.from_numpy(X[:5]).cuda(), 6)


def test_mirror_with_device_kmeans_backend():
    import multimodal_fusion_amd  # noqa: F401
    pp = import_module("multimodal_fusion_amd.build_hypergraph.preprocess_hypergraph")
    rng = np.random.RandomState(2)
    centers = rng.randn(4, 16).astype(np.float32) * 5.0
    X = torch.from_numpy(np.concatenate([c + rng.randn(50, 16).astype(np.float32) for c in centers], 0))
    prev = pp.KMEANS_BACKEND
    try:
        pp.set_kmeans_backend("sklearn")
        ei_s, ew_s, st_s = pp.build_hypergraph_knn_kmeans(X[:120], X[120:], None, 5, 4)
        pp.set_kmeans_backend("device")
        ei_d, ew_d, st_d = pp.build_hypergraph_knn_kmeans(X[:120], X[120:], None, 5, 4)
        ei_d2, ew_d2, _ = pp.build_hypergraph_knn_kmeans(X[:120], X[120:], None, 5, 4)
        assert torch.equal(ei_d, ei_d2) and torch.equal(ew_d, ew_d2)          # the device KMeans is deterministic
    finally:
        pp.set_kmeans_backend(prev)
    # separable blobs: both backends find the same 4 cliques, hence the same edge set and weights
    assert torch.equal(ei_s, ei_d) and torch.equal(ew_s, ew_d) and st_s == st_d


@pytest.mark.parametrize("n,d,R,group", [(1, 1, 1, 1), (130, 33, 7, 3), (1000, 512, 60, 6), (257, 1024, 20, 5), (300, 3000, 4, 2)])
def test_seed_distances_against_torch(n, d, R, group):
    """mmf_seed_distances: squared distances to the candidate rows, clamped per seeding — against float64 torch."""
    import multimodal_fusion_amd as mmf
    g = torch.Generator().manual_seed(n + d)
    X = torch.randn(n, d, generator=g).cuda()
    cand = torch.randint(0, n, (R,), generator=g).cuda()
    closest = torch.rand(-(-R // group), n, generator=g).cuda() * d
    ref = ((X[cand].double()[:, None, :] - X.double()[None, :, :]) ** 2).sum(-1)
    out = mmf.ops.seed_distances(X, cand)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    out2 = mmf.ops.seed_distances(X, cand, group, closest)
    ref2 = torch.minimum(ref, closest.double()[torch.arange(R, device="cuda") // group])
    torch.testing.assert_close(out2.double(), ref2, rtol=1e-5, atol=1e-5)
    assert torch.equal(out2, mmf.ops.seed_distances(X, cand, group, closest))      # deterministic
    rows = torch.randn(R, d, generator=g).cuda()                                   # explicit candidate rows (centroids)
    out3 = mmf.ops.seed_distances(X, rows)
    ref3 = ((rows.double()[:, None, :] - X.double()[None, :, :]) ** 2).sum(-1)
    torch.testing.assert_close(out3.double(), ref3, rtol=1e-5, atol=1e-5)


def test_kmeanspp_seed_follows_its_definition():
    """mmf_kmeanspp_seed against a float64 torch transcription driven by the same uniforms: same centres, step by step
    (inverse-CDF draws proportional to the running closest-centre distance, best of `trials` by potential)."""
    import multimodal_fusion_amd as mmf
    g = torch.Generator().manual_seed(3)
    n, d, k, n_init, trials = 700, 24, 9, 4, 3
    X = torch.randn(n, d, generator=g).cuda()
    u0 = torch.rand(n_init, generator=g).cuda()
    us = torch.rand(k - 1, n_init, trials, generator=g).cuda()
    got = mmf.ops.kmeanspp_seed(X, k, u0, us)
    assert torch.equal(got, mmf.ops.kmeanspp_seed(X, k, u0, us))                   # deterministic
    Xd = X.double()
    for i in range(n_init):
        first = min(int(float(u0[i]) * n), n - 1)
        assert int(got[i, 0]) == first
        closest = ((Xd - Xd[first]) ** 2).sum(1)
        for s in range(1, k):
            cs = torch.cumsum(closest, 0)
            best, best_pot, best_row = None, None, None
            for t in range(trials):
                target = float(us[s - 1, i, t]) * float(cs[-1])
                j = min(int(torch.searchsorted(cs, torch.tensor(target, dtype=torch.float64, device="cuda"))), n - 1)
                row = torch.minimum(closest, ((Xd - Xd[j]) ** 2).sum(1))
                pot = float(row.sum())
                if best is None or pot < best_pot:
                    best, best_pot, best_row = j, pot, row
            # a draw that lands within rounding of a boundary may pick a neighbour: accept the kernel's pick if its
            # potential is the same to f32 accuracy
            gi = int(got[i, s])
            if gi != best:
                alt = torch.minimum(closest, ((Xd - Xd[gi]) ** 2).sum(1))
                assert abs(float(alt.sum()) - best_pot) <= 1e-4 * best_pot, (i, s, gi, best)
                best_row = alt
            closest = best_row
