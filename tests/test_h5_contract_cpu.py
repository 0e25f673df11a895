"""HDF5 layout contract of the pipelines through the test-owned in-memory store (tests/memh5.py):
group / dataset names, dtypes, [2, E] int64 contiguity, attrs, overwrite behaviour (Appendix A4), JSON-able stats (A3),
loader fallbacks.  No GPU and no h5py needed: these functions only move arrays."""
import json
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from memh5 import MemStore   # noqa: E402


@pytest.fixture()
def env():
    b = import_module("multimodal_fusion_amd.build_hypergraph")
    store = MemStore()
    b.h5io.set_file_opener(store)
    yield b, store
    b.h5io.set_file_opener(None)


def test_loaders_and_their_fallbacks(env):
    b, store = env
    F = np.random.RandomState(0).randn(7, 5).astype(np.float64)       # loaders force float32 (:50)
    store.new_case("a.h5", F, np.arange(14).reshape(7, 2), np.ones((3, 5)))
    wf, wp = b.load_wsi_data("a.h5")
    assert wf.dtype == torch.float32 and wp.dtype == torch.float32 and tuple(wf.shape) == (7, 5) and tuple(wp.shape) == (7, 2)
    assert torch.equal(wf, torch.from_numpy(F).float())
    assert tuple(b.load_tma_data("a.h5").shape) == (3, 5)
    store.new_case("b.h5", F)                                          # no positions, no tma
    wf, wp = b.load_wsi_data("b.h5")
    assert tuple(wp.shape) == (7, 2) and float(wp.abs().sum()) == 0.0   # dummy zero positions (:57-60)
    assert b.load_tma_data("b.h5") is None
    with store("c.h5", "a") as f:
        f.create_group("wsi")
    with pytest.raises(ValueError, match="WSI features not found in c.h5"):
        b.load_wsi_data("c.h5")
    assert b.load_similarity_matrices("b.h5") == (None, None)


def test_writer_layout_dtypes_attrs_and_overwrite(env):
    b, store = env
    store.new_case("p.h5", np.zeros((4, 3), np.float32))
    sf, sp, tf = torch.rand(5, 3), torch.rand(5, 2), torch.rand(2, 3)
    ei = torch.tensor([[0, 0, 1], [1, 2, 2]], dtype=torch.int64).t().contiguous().t()      # a non-contiguous [2, 3] view
    assert not ei.is_contiguous()
    ew = torch.tensor([0.5, 0.25, 1.0])
    stats = {"grouping": {"group_sizes": [np.int64(3), np.int64(2)]}, "x": np.float32(0.5), "n": 3}   # A3: numpy scalars
    Kw, Swt = torch.rand(4, 4), torch.rand(5, 2)
    for rep in range(2):                                                # second call overwrites (A4) instead of raising
        b.save_hypergraph_to_h5("p.h5", sf, sp, tf, ei, ew, np.array([0, 1, 0, 1, 1], dtype=np.int32), stats,
                                wsi_similarity_matrix=Kw, wsi_tma_similarity_matrix=Swt)
    f = store.files["p.h5"]
    hg = f["hypergraph"]
    assert set(hg.keys()) == {"wsi_super", "tma", "edge_index", "edge_weights", "group_labels", "similarity"}
    assert set(hg["wsi_super"].keys()) == {"features", "positions"} and set(hg["tma"].keys()) == {"features"}
    assert set(hg["similarity"].keys()) == {"wsi_internal", "wsi_tma"}
    e = hg["edge_index"][:]
    assert e.dtype == np.int64 and e.shape == (2, 3) and e.flags["C_CONTIGUOUS"] and np.array_equal(e, [[0, 0, 1], [1, 2, 2]])
    assert hg["edge_weights"][:].dtype == np.float32 and hg["wsi_super"]["features"][:].dtype == np.float32
    assert hg["group_labels"][:].dtype == np.int32
    assert list(hg["similarity"].attrs["wsi_shape"]) == [4, 4] and list(hg["similarity"].attrs["wsi_tma_shape"]) == [5, 2]
    back = json.loads(hg.attrs["stats"])
    assert back == {"grouping": {"group_sizes": [3, 2]}, "x": 0.5, "n": 3}
    a, c = b.load_similarity_matrices("p.h5")
    assert torch.equal(a, Kw) and torch.equal(c, Swt)
    assert "wsi" in f and "features" in f["wsi"]                        # inputs untouched
    # the reader of downstream_survival/datasets/multimodal_dataset.py:342-386 addresses these four paths
    for path in ("hypergraph/wsi_super/features", "hypergraph/tma/features", "hypergraph/edge_index", "hypergraph/edge_weights"):
        assert f[path][:] is not None
    # without the optional matrices no similarity group appears
    store.new_case("q.h5", np.zeros((4, 3), np.float32))
    b.save_hypergraph_to_h5("q.h5", sf, sp, tf, ei, ew, np.zeros(5, np.int64), {})
    assert "similarity" not in store.files["q.h5"]["hypergraph"]


def test_dataset_shell_skips_missing_and_failing_files(env, tmp_path, monkeypatch):
    b, store = env
    pp = import_module("multimodal_fusion_amd.build_hypergraph.preprocess_hypergraph")
    store.new_case(os.path.join("root", "ok.h5"), np.zeros((4, 3), np.float32))
    store.new_case(os.path.join("root", "bad.h5"), np.zeros((4, 3), np.float32))
    csv = tmp_path / "cases.csv"
    csv.write_text("case_id,h5_file_path\nA,ok.h5\nB,missing.h5\nC,bad.h5\n")
    nocol = tmp_path / "nocol.csv"
    nocol.write_text("case_id,path\nA,ok.h5\n")
    with pytest.raises(ValueError, match="h5_file_path"):
        b.process_dataset(str(nocol), "root")

    def fake_single(path, *a, **k):
        if path.endswith("bad.h5"):
            raise RuntimeError("boom")
        return {"status": "ok", "n": np.int64(3)}
    monkeypatch.setattr(pp, "process_single_file", fake_single)
    monkeypatch.setattr(pp, "rebuild_hypergraph_from_similarity", fake_single)
    out = tmp_path / "stats.json"
    for fn in (b.process_dataset, b.batch_rebuild_hypergraph):
        res = fn(str(csv), "root", output_stats_path=str(out))
        assert [r["case_id"] for r in res] == ["A"] and res[0]["h5_path"] == "ok.h5"     # missing skipped, failing reported and skipped
        assert json.load(open(out)) == [{"status": "ok", "n": 3, "case_id": "A", "h5_path": "ok.h5"}]


def test_h5py_is_only_needed_when_a_file_is_opened():
    b = import_module("multimodal_fusion_amd.build_hypergraph")
    b.h5io.set_file_opener(None)
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed here")
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            b.load_wsi_data("nowhere.h5")


def test_written_hypergraph_reads_back_through_the_consumers_channels(env):
    """What save_hypergraph_to_h5 writes is what the training data set's `hypergraph=<key>` channels read
    (downstream_survival/datasets/multimodal_dataset.py:342-386, restated in h5io.read_hypergraph_channels): names, dtypes,
    shapes, the [2, E] int64 index, the fallbacks to the raw wsi / tma features."""
    b, store = env
    rng = np.random.RandomState(1)
    store.new_case("q.h5", rng.randn(9, 4).astype(np.float32), rng.rand(9, 2).astype(np.float32), rng.randn(3, 4).astype(np.float32))
    sf, sp, tf = torch.rand(5, 4), torch.rand(5, 2), torch.rand(3, 4)
    ei = torch.tensor([[0, 0, 1, 2], [1, 2, 2, 7]], dtype=torch.int64)
    ew = torch.tensor([0.5, 0.25, 1.0, 0.0])
    b.save_hypergraph_to_h5("q.h5", sf, sp, tf, ei, ew, np.zeros(5, np.int32), {"k": 1})
    ch = b.h5io.read_hypergraph_channels("q.h5")
    assert set(ch) == set(b.h5io.HYPERGRAPH_CHANNELS)
    assert torch.equal(ch["hypergraph=wsi_super_features"], sf) and ch["hypergraph=wsi_super_features"].dtype == torch.float32
    assert torch.equal(ch["hypergraph=tma_features"], tf)
    assert torch.equal(ch["hypergraph=edge_index"], ei) and ch["hypergraph=edge_index"].dtype == torch.int64
    assert torch.equal(ch["hypergraph=edge_weights"], ew[None, :])            # 1-D -> [1, E], as upstream's _standardize_array
    # a file whose hypergraph group lacks the feature copies: the reader falls back to the raw features (:353-369)
    with store("q.h5", "a") as f:
        del f["hypergraph"]["wsi_super"]
        del f["hypergraph"]["tma"]
    ch = b.h5io.read_hypergraph_channels("q.h5", ("hypergraph=wsi_super_features", "hypergraph=tma_features"))
    assert tuple(ch["hypergraph=wsi_super_features"].shape) == (9, 4) and tuple(ch["hypergraph=tma_features"].shape) == (3, 4)
    store.new_case("r.h5", rng.randn(2, 4).astype(np.float32))
    with pytest.raises(AssertionError, match="Hypergraph data not found"):
        b.h5io.read_hypergraph_channels("r.h5")
