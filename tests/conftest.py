"""pytest configuration: the `gpu` marker and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def unit_rows(n, d, seed):
    """Same synthetic generator as bench.py / SURVEY.md §8(d): randn rows, L2-normalised, f32."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)


@pytest.fixture(params=["device", "sklearn"])
def kmeans_backend(request):
    """Run a mirror test once per KMeans backend: the default device KMeans (csrc/mmf_kmeans.hip: scikit-learn's fit,
    decision for decision) and the reference's own scikit-learn call on the host.  Both must reproduce the fixtures the
    reference produced (labels, and the edges built on them)."""
    from importlib import import_module
    import multimodal_fusion_amd  # noqa: F401
    pp = import_module("multimodal_fusion_amd.build_hypergraph.preprocess_hypergraph")
    prev = pp.KMEANS_BACKEND
    pp.set_kmeans_backend(request.param)
    yield request.param
    pp.set_kmeans_backend(prev)
