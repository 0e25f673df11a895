"""A minimal in-memory stand-in for the part of the h5py File / Group protocol the hypergraph pipelines use
(`in`, `[]`, `[] =`, `del`, `create_group`, `.attrs`, dataset `[:]`, context manager).  TEST-OWNED: h5py is not in the
image, so the layout contract of build_hypergraph/h5io.py is exercised through this object; on-disk HDF5 bytes stay
"parity unpinned".  It is deliberately as strict as h5py where the reference trips over it: assigning to a name
that exists raises, writing through a file opened 'r' raises."""
import numpy as np


class MemDataset:
    def __init__(self, value):
        self._a = np.array(value)          # h5py stores a copy

    def __getitem__(self, key):
        return self._a[key].copy() if isinstance(self._a[key], np.ndarray) else self._a[key]

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype


class MemAttrs(dict):
    def __init__(self, owner):
        super().__init__()
        self._owner = owner

    def __setitem__(self, key, value):
        self._owner._check_writable()
        if isinstance(value, (list, tuple)):
            value = np.asarray(value)      # what h5py hands back for a list attribute
        elif not isinstance(value, (str, bytes, np.ndarray, int, float, np.generic)):
            raise TypeError(f"attribute {key!r}: unsupported type {type(value).__name__}")
        super().__setitem__(key, value)


class MemGroup:
    def __init__(self, root=None):
        self._items = {}
        self._root = root if root is not None else self
        self.attrs = MemAttrs(self)

    def _check_writable(self):
        if self._root._mode == "r":
            raise OSError("file is opened read-only")

    def __contains__(self, name):
        return name in self._items

    def __getitem__(self, name):
        node = self
        for part in name.strip("/").split("/"):
            node = node._items[part]
        return node

    def __setitem__(self, name, value):
        self._check_writable()
        if name in self._items:
            raise OSError(f"Unable to create link (name already exists): {name}")
        self._items[name] = MemDataset(value)

    def __delitem__(self, name):
        self._check_writable()
        del self._items[name]

    def create_group(self, name):
        self._check_writable()
        if name in self._items:
            raise ValueError(f"Unable to create group (name already exists): {name}")
        g = MemGroup(self._root)
        self._items[name] = g
        return g

    def keys(self):
        return self._items.keys()


class MemFile(MemGroup):
    def __init__(self):
        super().__init__(None)
        self._mode = "a"

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._mode = "closed-" + self._mode
        return False


class MemStore:
    """opener for build_hypergraph.h5io.set_file_opener: path -> MemFile, with h5py's mode semantics."""

    def __init__(self):
        self.files = {}
        self.opens = []

    def exists(self, path):
        return path in self.files

    def __call__(self, path, mode):
        self.opens.append((path, mode))
        if mode == "r":
            if path not in self.files:
                raise FileNotFoundError(f"Unable to open file (no such file): {path}")
        elif mode == "a":
            self.files.setdefault(path, MemFile())
        else:
            raise ValueError(f"mode {mode!r} is not used by the pipelines")
        f = self.files[path]
        f._mode = mode
        return f

    def new_case(self, path, wsi_features, wsi_positions=None, tma_features=None):
        """A per-patient input file as the reference expects it (preprocess_hypergraph.py:47-84)."""
        with self(path, "a") as f:
            w = f.create_group("wsi")
            w["features"] = np.asarray(wsi_features)
            if wsi_positions is not None:
                w["positions"] = np.asarray(wsi_positions)
            if tma_features is not None:
                f.create_group("tma")["features"] = np.asarray(tma_features)
        return path
