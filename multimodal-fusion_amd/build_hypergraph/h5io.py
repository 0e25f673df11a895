"""The HDF5 side of the reference's pipelines, written against the h5py File / Group protocol.

Layout the reference reads and writes (build_hypergraph/README.md:58-73, preprocess_hypergraph.py:47-84, 474-511;
read back by downstream_survival/datasets/multimodal_dataset.py:342-386):

    wsi/features              [N_wsi, D]            input
    wsi/positions             [N_wsi, 2|3]          input, optional (absent -> zeros [N_wsi, 2])
    tma/features              [N_tma, D]            input, optional (absent -> the file is skipped)
    hypergraph/wsi_super/features | positions        f32
    hypergraph/tma/features                          f32
    hypergraph/edge_index     [2, E] int64, C-contiguous
    hypergraph/edge_weights   [E]    f32
    hypergraph/group_labels   [N_wsi_super]          integer
    hypergraph/similarity/wsi_internal [N_wsi, N_wsi] f32, attrs['wsi_shape']     = [N_wsi, N_wsi]
    hypergraph/similarity/wsi_tma      [S, N_tma]     f32, attrs['wsi_tma_shape'] = [S, N_tma]
    hypergraph.attrs['stats']                         JSON string

h5py is imported when a file is opened, not when this package is imported: the arithmetic mirrors stay usable on a
machine without it.  `set_file_opener` swaps the opener for any callable `(path, mode) -> file-like` that speaks the
same protocol (`in`, `[]`, `[] =`, `del`, `create_group`, `.attrs`, dataset `[:]`, context manager) — the tests use
an in-memory store of their own; on-disk HDF5 bytes are "parity unpinned" until h5py exists where the tests run.

Two deliberate repairs of reference defects: datasets that already exist are replaced instead
of raising "name already exists" (A4: the reference's rebuild cannot overwrite its own previous output), and numpy
scalars inside `stats` are converted so that `json.dumps` does not raise after the datasets were written (A3).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Optional

import numpy as np

_OPENER: Optional[Callable] = None


def set_file_opener(opener: Optional[Callable]) -> None:
    """opener(path, mode) -> an object with the h5py.File protocol; None restores h5py.  An opener may also provide
    `opener.exists(path) -> bool` for stores that are not files."""
    global _OPENER
    _OPENER = opener


def open_file(path: str, mode: str):
    if _OPENER is not None:
        return _OPENER(path, mode)
    try:
        import h5py
    except ImportError as e:
        raise ImportError("the HDF5 pipeline functions need h5py (not installed); install it, or hand "
                          "build_hypergraph.h5io.set_file_opener a compatible opener") from e
    return h5py.File(path, mode)


def path_exists(path: str) -> bool:
    ex = getattr(_OPENER, "exists", None) if _OPENER is not None else None
    return bool(ex(path)) if ex is not None else os.path.exists(path)


def put(group, name: str, array) -> None:
    """group[name] = array, replacing an existing dataset (Appendix A4)."""
    if name in group:
        del group[name]
    group[name] = array


def child(group, name: str):
    """The sub-group `name`, created when missing (preprocess_hypergraph.py:476-477, 482-483, 488-489, 499-500)."""
    if name not in group:
        group.create_group(name)
    return group[name]


def jsonable(obj):
    """Python scalars / lists all the way down (Appendix A3: np.int64 in stats['grouping']['group_sizes'])."""
    if isinstance(obj, dict):
        return {str(k): jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [jsonable(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return jsonable(obj.tolist())
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def dumps_stats(stats) -> str:
    return json.dumps(jsonable(stats))


# ---------------------------------------------------------------------------------------------------------------------
# the CONSUMER's view of the layout: what downstream_survival/ reads back
# ---------------------------------------------------------------------------------------------------------------------
HYPERGRAPH_CHANNELS = ("hypergraph=wsi_super_features", "hypergraph=tma_features", "hypergraph=edge_index",
                       "hypergraph=edge_weights")


def _standardize(arr) -> np.ndarray:
    """2-D float32, as multimodal_dataset.py:454-463 hands every non-index array to the model."""
    arr = np.asarray(arr)
    if arr.ndim == 1:
        arr = arr.reshape(1, arr.shape[0])
    elif arr.ndim > 2:
        arr = arr.reshape(arr.shape[0], -1)
    return arr.astype(np.float32, copy=False)


def read_hypergraph_channels(h5_path: str, channels=HYPERGRAPH_CHANNELS) -> dict:
    """The `hypergraph=<key>` channels of a processed file as the training data set reads them
    (downstream_survival/datasets/multimodal_dataset.py:342-386): `wsi_super_features` <- hypergraph/wsi_super/features (falls
    back to wsi/features), `tma_features` <- hypergraph/tma/features (falls back to tma/features), `edge_index` <-
    hypergraph/edge_index as int64, `edge_weights` <- hypergraph/edge_weights (optional).  Arrays other than edge_index come
    back as 2-D float32 torch tensors (a 1-D edge_weights vector becomes [1, E], as upstream).  This is the contract the writer
    (save_hypergraph_to_h5) is tested against; the data set class itself is outside this package's scope."""
    import torch
    out = {}
    with open_file(h5_path, "r") as f:
        for channel in channels:
            if not channel.startswith("hypergraph="):
                raise ValueError(f"not a hypergraph channel: {channel}")
            key = channel[len("hypergraph="):]
            if "hypergraph" not in f:
                raise AssertionError("Hypergraph data not found in h5 file")
            hg = f["hypergraph"]
            if key == "wsi_super_features":
                src = hg["wsi_super"]["features"] if ("wsi_super" in hg and "features" in hg["wsi_super"]) else f["wsi"]["features"]
                out[channel] = torch.from_numpy(_standardize(src[:]))
            elif key == "tma_features":
                src = hg["tma"]["features"] if ("tma" in hg and "features" in hg["tma"]) else f["tma"]["features"]
                out[channel] = torch.from_numpy(_standardize(src[:]))
            elif key == "edge_index":
                if "edge_index" not in hg:
                    raise AssertionError("Failed to read hypergraph edge_index")
                out[channel] = torch.from_numpy(np.asarray(hg["edge_index"][:])).long()
            elif key == "edge_weights":
                if "edge_weights" in hg:                      # optional upstream
                    out[channel] = torch.from_numpy(_standardize(hg["edge_weights"][:]))
            else:
                raise AssertionError(f"Unknown hypergraph key: {key}")
    return out
