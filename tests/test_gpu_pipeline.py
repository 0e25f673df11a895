"""The cluster-shaped kernels, the order statistics, the fused WSI x TMA similarity and the two HDF5 pipelines on the GPU.

Pipelines: the arithmetic of process_single_file / rebuild_hypergraph_from_similarity is compared with golden G8 — the
reference's own functions called in the pipelines' order (tests/golden/make_golden.py) — with the KMeans steps on
the reference's scikit-learn call; files go through the test-owned in-memory store (tests/memh5.py; h5py is not in the
image, on-disk bytes are parity-unpinned).  The default device-KMeans flow (BASELINE C1: quick_rebuild_example) is checked
end to end for layout, determinism and structural properties."""
import itertools
import json
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from memh5 import MemStore   # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def mmf():
    import multimodal_fusion_amd as m
    assert torch.cuda.is_available()
    return m


@pytest.fixture()
def bh_store():
    b = import_module("multimodal_fusion_amd.build_hypergraph")
    store = MemStore()
    b.h5io.set_file_opener(store)
    yield b, store
    b.h5io.set_file_opener(None)


# ------------------------------------------------------------------------------------------------ segments
@pytest.mark.parametrize("n,S,seed", [(1, 1, 0), (70, 3, 1), (1024, 40, 2), (1025, 7, 3), (5000, 300, 4), (70000, 100, 5)])
def test_segment_kernels_against_numpy(mmf, n, S, seed):
    ops = mmf.ops
    rng = np.random.RandomState(seed)
    lab = rng.randint(0, S, size=n)
    if S > 2:
        lab[lab == 1] = 0                                  # an empty segment
    d = 37
    X = rng.randn(n, d).astype(np.float32)
    seg = ops.segment_sort(torch.from_numpy(lab).cuda(), S)
    order = np.argsort(lab, kind="stable")
    counts = np.bincount(lab, minlength=S)
    assert np.array_equal(seg.order.cpu().numpy(), order) and np.array_equal(seg.counts.cpu().numpy(), counts)
    assert np.array_equal(seg.offsets.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))
    mean = ops.segment_mean(torch.from_numpy(X).cuda(), seg).cpu().numpy()
    for c in range(S):
        if counts[c]:
            np.testing.assert_allclose(mean[c], X[lab == c].astype(np.float64).mean(0), rtol=0, atol=2e-6)
        else:
            assert np.all(np.isnan(mean[c]))
    lo, hi = ops.clique_pairs(seg)
    if n <= 5000:
        ref = [p for c in range(S) for p in itertools.combinations(np.nonzero(lab == c)[0].tolist(), 2)]
        assert lo.numel() == len(ref)
        if ref:
            assert np.array_equal(np.stack([lo.cpu().numpy(), hi.cpu().numpy()], 1), np.array(ref, dtype=np.int64))
    else:
        assert lo.numel() == int((counts * (counts - 1) // 2).sum()) and bool((lo < hi).all())
        assert bool((torch.from_numpy(lab).cuda()[lo] == torch.from_numpy(lab).cuda()[hi]).all())
    if n <= 5000 and n > 1:
        K = rng.rand(n, n).astype(np.float32)
        got = ops.segment_offdiag_mean(torch.from_numpy(K).cuda(), seg).cpu().numpy()
        for c in range(S):
            idx = np.nonzero(lab == c)[0]
            if len(idx) > 1:
                sub = K[np.ix_(idx, idx)].astype(np.float64)
                ref_m = (sub.sum() - np.trace(sub)) / (len(idx) * (len(idx) - 1))
                assert abs(got[c] - ref_m) < 1e-12 * max(1.0, len(idx))
            else:
                assert np.isnan(got[c])
    with pytest.raises(ValueError, match="outside"):
        ops.segment_sort(torch.full((5,), S, dtype=torch.int64).cuda(), S)


def test_knn_pairs_is_the_undirected_dedup_of_the_reference(mmf):
    """preprocess_hypergraph.py:386-388 + :403: set(tuple(sorted(e))) over the directed k-NN pairs, minus what the
    cliques of `labels` already hold."""
    ops = mmf.ops
    X = torch.randn(600, 16, generator=torch.Generator().manual_seed(3)).cuda()
    nbr, _ = mmf.simtopk(X, metric="neg_sq_l2", k=6)
    lab = torch.randint(0, 9, (600,), generator=torch.Generator().manual_seed(4)).cuda()
    for labels in (None, lab):
        lo, hi = ops.knn_pairs(nbr, labels)
        got = sorted(zip(lo.cpu().tolist(), hi.cpu().tolist()))
        nb, lb = nbr.cpu().numpy(), (None if labels is None else labels.cpu().numpy())
        ref = {tuple(sorted((i, int(j)))) for i in range(600) for j in nb[i]}
        if lb is not None:
            ref = {e for e in ref if lb[e[0]] != lb[e[1]]}
        assert got == sorted(ref) and len(got) == len(set(got))


# ------------------------------------------------------------------------------------------------ order statistics
@pytest.mark.parametrize("count", [1, 2, 5, 4095, 4096, 4097, 65536 + 3, 3_000_001, 4_700_003])
def test_array_stats_and_lower_median_match_torch(mmf, count):
    ops = mmf.ops
    g = torch.Generator(device="cuda").manual_seed(count)
    for kind in ("spread", "clustered"):
        v = torch.rand(count, generator=g, device="cuda")
        if kind == "clustered":
            v = 0.97 + 1e-4 * v                              # mean >> std: the pivot keeps the variance accurate
        med = ops.lower_median(v)
        assert float(med) == float(v.median())
        st = ops.array_stats(v)
        v64 = v.double()
        assert st["median"] == float(v.median()) and st["min"] == float(v.min()) and st["max"] == float(v.max())
        assert abs(st["mean"] - float(v64.mean())) <= 1e-6 * abs(float(v64.mean()))
        if count > 1:
            assert abs(st["std"] - float(v64.std())) <= 2e-6 * float(v64.std()) + 1e-12
        else:
            assert np.isnan(st["std"])
        assert repr(ops.array_stats(v)) == repr(st)          # bit-reproducible (repr: NaN compares unequal to itself)


@pytest.mark.parametrize("kind", ["uniform", "normal_tiny_range", "five_values", "constant", "bimodal", "sorted", "with_inf",
                                  "negative", "unaligned"])
def test_one_sweep_median_is_exact_and_falls_back(mmf, kind, monkeypatch):
    """Populations of 4 M values or more take the sampled-bracket path (one sweep, exact counts); whatever the sample
    says the result must be torch.median's element, and the four-pass radix select must agree (MMF_MEDIAN_RADIX=1)."""
    ops = mmf.ops
    count = 5_000_003
    g = torch.Generator(device="cuda").manual_seed(11)
    u = torch.rand(count + 1, generator=g, device="cuda")
    if kind == "uniform":
        v = u[:count]
    elif kind == "normal_tiny_range":
        v = (0.75 + 1e-6 * torch.randn(count, generator=g, device="cuda"))
    elif kind == "five_values":
        v = torch.floor(u[:count] * 5) * 0.125          # 20 % of the entries equal the median: the buffer overflows -> radix
    elif kind == "constant":
        v = torch.full((count,), 0.3125, device="cuda")
    elif kind == "bimodal":
        v = torch.where(u[:count] < 0.5, 0.1 * u[:count], 0.9 + 0.1 * u[:count])   # the median sits at the edge of a gap
    elif kind == "sorted":
        v = torch.sort(u[:count]).values
    elif kind == "with_inf":
        v = u[:count].clone(); v[::7] = float("inf"); v[3::11] = -float("inf")
    elif kind == "negative":
        v = -1e3 * u[:count] - 5.0
    else:
        v = u[1:count + 1]                               # 4-byte aligned only
    ref = float(v.median())
    assert float(ops.lower_median(v)) == ref, kind
    assert ops.array_stats(v)["median"] == ref
    monkeypatch.setenv("MMF_MEDIAN_RADIX", "1")
    assert float(ops.lower_median(v)) == ref
    monkeypatch.delenv("MMF_MEDIAN_RADIX")


@pytest.mark.parametrize("n", [2300, 2051])
def test_one_sweep_offdiag_median_on_structured_matrices(mmf, n):
    """Off-diagonal medians of matrices with strong row / column effects (the sample is uniform over ENTRIES, so it does
    not care) and a diagonal that would move the median if it were counted."""
    ops = mmf.ops
    g = torch.Generator(device="cuda").manual_seed(n)
    r = torch.rand(n, 1, generator=g, device="cuda") ** 3
    c = torch.rand(1, n, generator=g, device="cuda")
    K = (r * c + 0.01 * torch.rand(n, n, generator=g, device="cuda")).contiguous()
    K.fill_diagonal_(1e6)
    off = K[~torch.eye(n, dtype=torch.bool, device="cuda")]
    assert float(ops.offdiag_lower_median(K)) == float(off.median())
    K2 = torch.exp(-3.0 * torch.rand(n, n, generator=g, device="cuda"))
    K2 = torch.minimum(K2, K2.t()).contiguous()
    assert float(ops.offdiag_lower_median(K2)) == float(K2[~torch.eye(n, dtype=torch.bool, device="cuda")].median())


# ------------------------------------------------------------------------------------------------ a7 / f4
@pytest.mark.parametrize("N,M", [(8, 12), (100, 37)])
def test_wsi_tma_similarity_against_the_reference(mmf, N, M):
    bh = import_module("multimodal_fusion_amd.build_hypergraph")
    g = load_golden("g3_cross.npz")
    A, B = torch.from_numpy(g[f"N{N}_M{M}_A"]), torch.from_numpy(g[f"N{N}_M{M}_B"])
    for lam, lg in ((1.0, 1.0), (0.5, 3.0)):
        S, st = bh.compute_wsi_tma_similarity(A, torch.rand(N, 2), B, lam, lg)
        np.testing.assert_allclose(S.numpy(), g[f"N{N}_M{M}_S_lam{lam if lam != 1.0 else 1}"], rtol=0, atol=TOL)
        ref = g[f"N{N}_M{M}_stats_lam{lam if lam != 1.0 else 1}"]
        np.testing.assert_allclose([st[q] for q in ("mean", "std", "min", "max", "median")], ref, rtol=2e-5, atol=1e-6)
        json.dumps(st)


@pytest.mark.parametrize("n,m,d,dt", [(1, 1, 1, torch.float32), (130, 257, 70, torch.float32), (1000, 777, 128, torch.float32),
                                      (300, 300, 33, torch.float16), (2500, 1900, 96, torch.float32)])
def test_fused_direct_similarity_and_stats_against_the_oracle(mmf, n, m, d, dt):
    ops = mmf.ops
    g = torch.Generator().manual_seed(n + m)
    X = (torch.randn(n, d, generator=g) * 0.2).to(dt)
    Y = (torch.randn(m, d, generator=g) * 0.2).to(dt)
    S, st = ops.sim_dense_stats(X.cuda(), Y.cuda(), metric="rbf_direct", lam=0.8)
    ref = oracle.sim_dense(X.float().numpy(), Y.float().numpy(), metric="rbf_direct", lam=0.8)
    np.testing.assert_allclose(S.cpu().numpy(), ref, rtol=0, atol=TOL)
    assert torch.equal(S, ops.sim_dense(X.cuda(), Y.cuda(), metric="rbf_direct", lam=0.8))
    Sd = S.double()
    assert st["min"] == float(S.min()) and st["max"] == float(S.max()) and st["median"] == float(S.flatten().median())
    assert abs(st["mean"] - float(Sd.mean())) <= 1e-6 * float(Sd.mean())
    if n * m > 1:
        assert abs(st["std"] - float(Sd.std())) <= 1e-5 * float(Sd.std()) + 1e-9
    # nothing stored: rows recomputed in panels for each radix pass — the same five numbers, bit for bit
    none, st2 = ops.sim_dense_stats(X.cuda(), Y.cuda(), metric="rbf_direct", lam=0.8, store=False, panel_rows=256)
    assert none is None and repr(st2) == repr(st)
    # the matrix-core metrics go through the dense kernels + one reduction pass
    S3, st3 = ops.sim_dense_stats(X.cuda(), Y.cuda(), metric="rbf", lam=0.8)
    assert st3["median"] == float(S3.flatten().median()) and abs(st3["mean"] - float(S3.double().mean())) < 1e-6


# ------------------------------------------------------------------------------------------------ pipelines
def _edges_sorted(f):
    e = f["hypergraph/edge_index"][:]
    order = np.lexsort((e[1], e[0]))
    return e[:, order], f["hypergraph/edge_weights"][:][order]


def test_pipelines_reproduce_the_reference_arithmetic(bh_store, kmeans_backend):
    """process_single_file, then rebuild_hypergraph_from_similarity with other parameters and the edge-weight median
    filter, against golden G8."""
    b, store = bh_store
    g = load_golden("g8_pipeline.npz")
    S, G, k, H = (int(v) for v in g["params"])
    lam_h, lam_g = (float(v) for v in g["lambdas"])
    store.new_case("case.h5", g["wsi_features"], g["wsi_positions"], g["tma_features"])
    st = b.process_single_file("case.h5", S, G, k, H, lam_h, lam_g)
    f = store.files["case.h5"]
    np.testing.assert_allclose(f["hypergraph/wsi_super/features"][:], g["super_features"], rtol=0, atol=TOL)
    np.testing.assert_allclose(f["hypergraph/wsi_super/positions"][:], g["super_positions"], rtol=0, atol=TOL)
    np.testing.assert_allclose(f["hypergraph/similarity/wsi_internal"][:], g["K_wsi"], rtol=0, atol=TOL)
    np.testing.assert_allclose(f["hypergraph/similarity/wsi_tma"][:], g["sim"], rtol=0, atol=TOL)
    assert np.array_equal(f["hypergraph/tma/features"][:], g["tma_features"])
    assert np.array_equal(f["hypergraph/group_labels"][:], g["group_labels"])
    ei, ew = _edges_sorted(f)
    assert np.array_equal(ei, g["ei_sorted"]) and ei.dtype == np.int64
    np.testing.assert_allclose(ew, g["ew_sorted"], rtol=0, atol=TOL)
    assert json.loads(f["hypergraph"].attrs["stats"]) == json.loads(json.dumps(st))
    ws = st["wsi_aggregation"]
    got = [ws["avg_intra_cluster_similarity"]] + [ws["wsi_similarity_matrix_stats"][q] for q in ("mean", "std", "min", "max", "median")]
    np.testing.assert_allclose(got, g["wsi_stats"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose([st["similarity"][q] for q in ("mean", "std", "min", "max", "median")], g["sim_stats"], rtol=2e-5, atol=1e-6)
    assert st["grouping"]["group_sizes"] == g["group_sizes"].tolist() and st["hypergraph"]["num_edges"] == int(g["num_edges"])
    # ---- rebuild from the stored matrices (overwrites the group it reads from: Appendix A4) ----
    S2, G2 = (int(v) for v in g["rb_params"])
    ratio = float(g["rb_ratio"])
    st2 = b.rebuild_hypergraph_from_similarity("case.h5", S2, G2, k, H, ratio)
    np.testing.assert_allclose(f["hypergraph/wsi_super/features"][:], g["rb_super_features"], rtol=0, atol=TOL)
    np.testing.assert_allclose(f["hypergraph/similarity/wsi_tma"][:], g["rb_sim"], rtol=0, atol=TOL)
    np.testing.assert_allclose(f["hypergraph/similarity/wsi_internal"][:], g["K_wsi"], rtol=0, atol=TOL)     # kept as stored
    assert np.array_equal(f["hypergraph/group_labels"][:], g["rb_group_labels"])
    ei2, ew2 = _edges_sorted(f)
    assert np.array_equal(ei2, g["rb_ei_sorted"])
    np.testing.assert_allclose(ew2, g["rb_ew_sorted"], rtol=0, atol=TOL)
    h = st2["hypergraph"]
    assert h["num_edges"] == int(g["rb_num_edges"]) and h["num_edges_after_threshold"] == g["rb_ei_sorted"].shape[1]
    assert abs(h["threshold"] - float(g["rb_threshold"])) < 1e-6 and h["threshold_ratio"] == ratio
    np.testing.assert_allclose([st2["similarity"][q] for q in ("mean", "std", "min", "max", "median")], g["rb_sim_stats"], rtol=2e-5, atol=1e-6)
    # ---- rebuild keeping the stored super patches and groups (the None / None branch, :836-875) ----
    st3 = b.rebuild_hypergraph_from_similarity("case.h5", None, None, k, H)
    assert st3["grouping"] == {"method": "existing", "num_groups": int(len(np.unique(g["rb_group_labels"])))} and st3["wsi_aggregation"] == {}
    np.testing.assert_allclose([st3["similarity"][q] for q in ("mean", "std", "min", "max", "median")], g["rb_sim_stats"], rtol=2e-5, atol=1e-6)
    assert f["hypergraph/edge_index"][:].shape[1] == int(g["rb_num_edges"])                                  # no filter this time
    # a file without TMA features is skipped / refused exactly as upstream
    store.new_case("no_tma.h5", g["wsi_features"], g["wsi_positions"])
    assert b.process_single_file("no_tma.h5") == {"status": "skipped", "reason": "no_tma"}
    with pytest.raises(ValueError, match="TMA features not found"):
        b.rebuild_hypergraph_from_similarity("no_tma.h5")


def test_quick_rebuild_flow_on_the_device_kmeans(bh_store, tmp_path):
    """BASELINE C1's entry: process a small dataset, then quick_rebuild_example over it, default (device) KMeans."""
    b, store = bh_store
    qr = import_module("multimodal_fusion_amd.build_hypergraph.quick_rebuild_example")
    pp = import_module("multimodal_fusion_amd.build_hypergraph.preprocess_hypergraph")
    assert pp.KMEANS_BACKEND == os.environ.get("MMF_KMEANS_BACKEND", "device")
    rng = np.random.RandomState(0)
    rows = ["case_id,h5_file_path"]
    for c in range(2):
        cent = rng.randn(30, 48).astype(np.float32)
        W = (cent[rng.randint(0, 30, 4096)] * 0.3 + 0.04 * rng.randn(4096, 48)).astype(np.float32)
        T = (cent[rng.randint(0, 30, 64)] * 0.3 + 0.04 * rng.randn(64, 48)).astype(np.float32)
        store.new_case(os.path.join("root", f"c{c}.h5"), W, rng.rand(4096, 2).astype(np.float32), T)
        rows.append(f"P{c},c{c}.h5")
    rows.append("P9,absent.h5")
    csv = tmp_path / "cases.csv"
    csv.write_text("\n".join(rows) + "\n")
    res = b.process_dataset(str(csv), "root", num_wsi_super_patches=100, num_groups=10, hypergraph_k=5, num_hyperedges=10,
                            lambda_h=0.5, lambda_g=1.0, output_stats_path=str(tmp_path / "s.json"))
    assert [r["case_id"] for r in res] == ["P0", "P1"]
    json.load(open(tmp_path / "s.json"))
    snap = {}
    for rep in range(2):                                               # twice: overwrite works and results are deterministic
        out = qr.main(["--csv_path", str(csv), "--data_root_dir", "root", "--num_wsi_super_patches", "64", "--num_groups", "5",
                       "--threshold_median_ratio", "0.9", "--output_stats", str(tmp_path / "r.json")])
        assert len(out) == 2
        for c in range(2):
            f = store.files[os.path.join("root", f"c{c}.h5")]
            e, w = f["hypergraph/edge_index"][:], f["hypergraph/edge_weights"][:]
            assert e.dtype == np.int64 and e.shape[0] == 2 and e.flags["C_CONTIGUOUS"] and w.dtype == np.float32 and w.shape == (e.shape[1],)
            assert f["hypergraph/wsi_super/features"][:].shape == (64, 48) and f["hypergraph/similarity/wsi_tma"][:].shape == (64, 64)
            assert f["hypergraph/similarity/wsi_internal"][:].shape == (4096, 4096) and f["hypergraph/group_labels"][:].shape == (64,)
            assert np.all(e[0] < e[1]) and e.max() < 128 and np.all(np.diff(e[0] * 128 + e[1]) > 0)          # sorted, unique, undirected
            assert np.all((w >= 0) & (w <= 1 + 1e-6))
            st = json.loads(f["hypergraph"].attrs["stats"])
            assert st["hypergraph"]["num_edges_after_threshold"] == e.shape[1] <= st["hypergraph"]["num_edges"]
            key = (c, "e"), (c, "w")
            if rep == 0:
                snap[key[0]], snap[key[1]] = e.copy(), w.copy()
            else:
                assert np.array_equal(snap[key[0]], e) and np.array_equal(snap[key[1]], w)
