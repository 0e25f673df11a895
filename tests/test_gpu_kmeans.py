"""Device KMeans (csrc/mmf_kmeans.hip) against the call the reference makes,
``sklearn.cluster.KMeans(n_clusters, random_state=42, n_init=10).fit_predict`` (preprocess_hypergraph.py:150-151,
299-300, 391-392): the LABELS must be scikit-learn's, not just the objective.

Three layers of evidence:
  * scikit-learn run in the test on the same data (the GPU box's host cores);
  * the golden fixtures g5 / g8 / g9: scikit-learn's labels as the reference's own functions produced them in the build
    container (g9: the pipeline's real shape, N = 16384, d = 512, k = 100);
  * oracle/kmeans_restate.py, the CPU restatement of the contract (float64 sums, float32 roundings where scikit-learn
    stores float32): seeds, iteration counts and labels, restart by restart.
scikit-learn forms its float32 sums through BLAS in a machine-dependent order, so on data without structure a decision
that hangs on its rounding noise may differ between machines; the g9 "gauss" case documents that (see its test)."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def _km():
    import multimodal_fusion_amd  # noqa: F401
    return import_module("multimodal_fusion_amd.kmeans")


def _data(kind, n, d, rng):
    if kind == "gauss":
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == "blobs":
        c = rng.standard_normal((max(2, n // 40), d)).astype(np.float32) * 2
        return (c[rng.integers(0, len(c), n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    if kind == "unit":
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x / np.linalg.norm(x, axis=1, keepdims=True)
    a = rng.standard_normal((n, 16)).astype(np.float32)          # rows of an RBF similarity matrix (group_by_similarity)
    b = rng.standard_normal((d, 16)).astype(np.float32)
    return np.exp(-0.05 * ((a[:, None, :] - b[None]) ** 2).sum(-1)).astype(np.float32)


CASES = [("gauss", 4000, 64, 25), ("blobs", 4000, 64, 6), ("unit", 600, 128, 10), ("sim", 600, 64, 10), ("gauss", 1500, 16, 40),
         ("blobs", 1500, 64, 25), ("sim", 64, 64, 1), ("gauss", 40, 64, 3), ("blobs", 40, 16, 10), ("unit", 64, 33, 6),
         ("sim", 1500, 16, 40), ("unit", 4000, 128, 10), ("blobs", 160, 16, 40), ("gauss", 257, 1000, 7),
         ("gauss", 12, 5, 12), ("gauss", 300, 1, 4), ("unit", 65, 2, 64), ("gauss", 1, 7, 1), ("blobs", 3000, 31, 100)]


@pytest.mark.parametrize("kind,n,d,k", CASES)
def test_labels_are_sklearns(kind, n, d, k):
    """Labels identical to scikit-learn's own run AND to the CPU restatement; against the restatement also the seeds
    and the iteration count of every restart."""
    import multimodal_fusion_amd as mmf
    from sklearn.cluster import KMeans
    from oracle import kmeans_restate as kr
    km = _km()
    X = _data(kind, n, d, np.random.default_rng(1000 * n + d + k))
    ref = KMeans(n_clusters=k, random_state=42, n_init=10).fit(X)
    Xg = torch.from_numpy(X).cuda()
    first, u = km.sklearn_stream(42, 10, k, n)
    labels, centres, info, seeds = mmf.ops.kmeans_fit(Xg, k, first, u, return_seeds=True)
    ri = {}
    rl = kr.kmeans_fit_predict(X, k, info=ri)
    lab = labels.cpu().numpy()
    assert labels.dtype == torch.int64 and centres.shape == (k, d)
    assert np.array_equal(lab, rl), f"labels differ from the restatement in {int((lab != rl).sum())} places"
    for i in range(10):
        assert np.array_equal(seeds[i].cpu().numpy(), ri["per_init"][i]["seeds"]), f"restart {i}: seeding differs"
    assert info["best_init"] == ri["best_init"]
    # d = 1: numpy sums a contiguous column pairwise, the device row by row — the column mean, hence the centred data, may differ
    # by an ulp there (labels unaffected in every case tried); for d >= 2 both accumulate row by row and the data are identical
    assert abs(info["inertia"] - ri["inertia"]) <= (1e-6 if d == 1 else 1e-9) * ri["inertia"]
    assert np.array_equal(lab, ref.labels_), (f"labels differ from scikit-learn's in {int((lab != ref.labels_).sum())} places "
                                              f"(inertia {info['inertia']} vs {ref.inertia_}; ambiguous seeding decisions: "
                                              f"{info['ambiguous_draws']} draws, {info['ambiguous_trials']} trials)")
    np.testing.assert_allclose(centres.cpu().numpy(), ref.cluster_centers_, rtol=0, atol=2e-6 * max(1.0, float(np.abs(X).max())))
    assert abs(info["inertia"] - ref.inertia_) <= 1e-5 * ref.inertia_
    # deterministic, and the public wrapper returns the same thing
    l2, c2, inertia = km.kmeans_fit_predict(Xg, k)
    assert torch.equal(l2, labels) and torch.equal(c2, centres) and inertia == info["inertia"]


@pytest.mark.parametrize("tag", ["small", "mid", "zero"])
def test_golden_g5_labels(tag):
    """The labels the reference's build_hypergraph_knn_kmeans computed in the build container (golden G5)."""
    km = _km()
    g = load_golden("g5_knn_kmeans.npz")
    X = torch.from_numpy(np.concatenate([g[f"{tag}_W"], g[f"{tag}_T"]], 0)).cuda()
    labels, _, _ = km.kmeans_fit_predict(X, int(g[f"{tag}_H"]))
    assert np.array_equal(labels.cpu().numpy(), g[f"{tag}_labels"])


def test_pipeline_shape_against_golden_g9():
    """N = 16384, d = 512, k = 100 (the super-patch clustering of process_single_file, :516, :566): device labels against
    scikit-learn's labels from the build container (golden G9) and against the CPU restatement's seeds / iterations /
    labels, for 'clustered' rows (what patch embeddings look like) and for 'gauss' rows (no structure at all: 22-45 Lloyd
    iterations per restart, the least stable case).  scripts/kmeans_parity.py --big also runs scikit-learn on the GPU box's
    own host cores on the same data (its BLAS sums differ from the build container's: DESIGN.md §4.5)."""
    sys.path.insert(0, GOLDEN)
    from make_g9_kmeans import g9_data
    import multimodal_fusion_amd as mmf
    km = _km()
    g = load_golden("g9_kmeans_scale.npz")
    first, u = km.sklearn_stream(42, 10, 100, 16384)
    for kind in ("clustered", "gauss"):
        X = g9_data(kind)
        labels, _, info, seeds = mmf.ops.kmeans_fit(torch.from_numpy(X).cuda(), 100, first, u, return_seeds=True)
        lab = labels.cpu().numpy()
        assert np.array_equal(seeds.cpu().numpy(), g[f"{kind}_restate_seeds"]), f"{kind}: seeding differs from the restatement"
        assert info["n_iter_per_init"] == g[f"{kind}_restate_n_iter"].tolist(), f"{kind}: iteration counts differ from the restatement"
        assert info["best_init"] == int(g[f"{kind}_restate_best"])
        assert np.array_equal(lab, g[f"{kind}_restate_labels"]), f"{kind}: labels differ from the restatement"
        sk = g[f"{kind}_sklearn_labels"]           # scikit-learn's labels as the build container produced them: a fixed record
        assert np.array_equal(lab, sk), f"{kind}: {int((lab != sk).sum())} labels differ from scikit-learn's (golden g9)"
        assert abs(info["inertia"] - float(g[f"{kind}_sklearn_inertia"])) <= 1e-5 * info["inertia"]


def test_fewer_distinct_rows_than_clusters_and_errors():
    """Duplicate rows with n_distinct < k: empty clusters appear and scikit-learn's relocation rule runs; the fit must
    terminate, be deterministic and leave every distinct row in one cluster.  Bad arguments raise like scikit-learn."""
    km = _km()
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6, 24)).astype(np.float32)
    X = torch.from_numpy(base[rng.integers(0, 6, 200)]).cuda()
    l1, c1, i1 = km.kmeans_fit_predict(X, 9)
    l2, c2, i2 = km.kmeans_fit_predict(X, 9)
    assert torch.equal(l1, l2) and i1 == i2 and i1 <= 1e-8
    lab = l1.cpu().numpy()
    Xn = X.cpu().numpy()
    for r in range(6):
        rows = np.where((Xn == base[r]).all(1))[0]
        assert len(np.unique(lab[rows])) == 1
    assert len(np.unique(lab)) == 6
    with pytest.raises(ValueError, match="n_samples=5 should be >= n_clusters=6"):
        km.kmeans_fit_predict(X[:5], 6)
    with pytest.raises(RuntimeError):
        km.kmeans_fit_predict(X.cpu(), 3)


def test_relocation_of_an_empty_cluster_follows_sklearn():
    """Lattice data with many duplicate rows and k close to the number of distinct rows: clusters empty during Lloyd and
    scikit-learn's relocation (_relocate_empty_clusters_dense: the point farthest from its centre becomes the new centre
    and leaves its donor's sum) runs with a non-zero distance.  The seeds below are cases in which the CPU restatement
    relocates at least once (found by search); labels must equal scikit-learn's and the restatement's."""
    import warnings
    from sklearn.cluster import KMeans
    from oracle import kmeans_restate as kr
    km = _km()
    for seed in (27, 275, 288, 318, 424, 473, 482, 546):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([20, 30, 50]))
        k = int(rng.choice([6, 8, 12]))
        X = rng.integers(0, 4, (n, 2)).astype(np.float32) + rng.standard_normal((n, 2)).astype(np.float32) * float(rng.choice([0, 0.01, 0.3]))
        assert (n, k) == (20, 12)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = KMeans(n_clusters=k, random_state=42, n_init=10).fit_predict(X)
        lab, _, _ = km.kmeans_fit_predict(torch.from_numpy(X).cuda(), k)
        assert np.array_equal(lab.cpu().numpy(), kr.kmeans_fit_predict(X, k)), seed
        assert np.array_equal(lab.cpu().numpy(), ref), seed


def test_mirror_backends_agree():
    import multimodal_fusion_amd  # noqa: F401
    pp = import_module("multimodal_fusion_amd.build_hypergraph.preprocess_hypergraph")
    rng = np.random.RandomState(2)
    X = torch.from_numpy(rng.randn(200, 16).astype(np.float32))            # no structure: labels still have to agree
    prev = pp.KMEANS_BACKEND
    try:
        pp.set_kmeans_backend("sklearn")
        ei_s, ew_s, st_s = pp.build_hypergraph_knn_kmeans(X[:120], X[120:], None, 5, 4)
        pp.set_kmeans_backend("device")
        ei_d, ew_d, st_d = pp.build_hypergraph_knn_kmeans(X[:120], X[120:], None, 5, 4)
    finally:
        pp.set_kmeans_backend(prev)
    assert torch.equal(ei_s, ei_d) and torch.equal(ew_s, ew_d) and st_s == st_d
