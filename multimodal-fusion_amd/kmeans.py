"""Device KMeans: the reference's clustering call, reproduced on the GPU.

The reference clusters with ``sklearn.cluster.KMeans(n_clusters, random_state=42, n_init=10).fit_predict(x)`` on the host
(build_hypergraph/preprocess_hypergraph.py:150-151, 299-300, 391-392); the labels decide the super-patches, the groups
and the clique hyperedges.  ``mmf_kmeans_fit`` (csrc/mmf_kmeans.hip) takes the same decisions scikit-learn takes, in the
same order: k-means++ seeding with ``2 + int(log k)`` local trials, Lloyd iterations with scikit-learn's two convergence
tests, the best of ``n_init`` restarts by inertia — every inner product and sum in float64 on the matrix cores, rounded to
float32 exactly where scikit-learn stores a float32.  What makes the labels scikit-learn's own is its random stream, and
that stream does not depend on the data: this module draws it on the host with ``numpy.random.RandomState(seed)`` in
scikit-learn's order of consumption (sklearn/cluster/_kmeans.py 1.7.2, ``_kmeans_plusplus``: per restart one
``random_sample()`` for the first centre, then one ``uniform(size=trials)`` per seeding step) and hands it to the
library, which runs all restarts in lockstep.

scikit-learn's own float32 sums go through BLAS in an order that depends on the CPU and the thread count, so its labels
are themselves reproducible only up to the decisions that rounding noise takes; ``info['ambiguous_draws']`` /
``['ambiguous_trials']`` count the seeding decisions that came within 4 float32 ulps of going the other way.
"""
from __future__ import annotations

import functools
import math
from typing import Tuple

import numpy as np
import torch

from . import ops


@functools.lru_cache(maxsize=4)
def _uniform_cdf(n_samples: int) -> np.ndarray:
    """cdf of RandomState.choice(n, p=ones(n, float32) / float32(n)): cumsum of the float32 weights in float64, normalised."""
    p = (np.ones(n_samples, np.float32) / np.float32(n_samples)).astype(np.float64)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    cdf.setflags(write=False)
    return cdf


def sklearn_stream(seed: int, n_init: int, n_clusters: int, n_samples: int) -> Tuple[np.ndarray, np.ndarray]:
    """scikit-learn's draws for KMeans(n_clusters, random_state=seed, n_init=n_init) on n_samples points:
    (first centre of every restart, int64 [n_init]; uniforms float64 [n_init, n_clusters - 1, trials]).

    The first centre is ``RandomState.choice(n, p=ones(n, float32) / float32(n))``: one ``random_sample()`` looked up
    (side='right') in the normalised float64 cumulative sum of p (numpy/random/mtrand.pyx)."""
    trials = 2 + int(math.log(n_clusters))
    rs = np.random.RandomState(seed)
    cdf = _uniform_cdf(n_samples)
    # per restart the stream gives one double for the first centre, then `trials` doubles per seeding step (uniform(size=trials) is
    # 0.0 + 1.0 * the next doubles, bit for bit): ONE random_sample call yields the same doubles as scikit-learn's n_init * k calls
    # (oracle/kmeans_restate.py keeps the call-by-call form; the GPU tests compare the seeds of both)
    steps = max(n_clusters - 1, 0)
    r = rs.random_sample(n_init * (1 + steps * trials)).reshape(n_init, 1 + steps * trials)
    first = cdf.searchsorted(r[:, 0], side="right").astype(np.int64)
    u = np.ascontiguousarray(r[:, 1:]).reshape(n_init, steps, trials)
    np.clip(first, 0, n_samples - 1, out=first)
    return first, u


def kmeans_fit_predict(X: torch.Tensor, n_clusters: int, *, n_init: int = 10, max_iter: int = 300,
                       tol: float = 1e-4, seed: int = 42, return_info: bool = False):
    """Returns (labels int64 [N], centers f32 [k, D], inertia) — the labels of
    ``KMeans(n_clusters, random_state=seed, n_init=n_init, max_iter=max_iter, tol=tol).fit_predict(X)``.
    X must live on a ROCm device."""
    if not X.is_cuda:
        raise RuntimeError("kmeans_fit_predict: X must be on a ROCm device (no CPU path)")
    X = X.detach().float().contiguous()
    n = X.shape[0]
    if not (1 <= n_clusters <= n):
        raise ValueError(f"n_samples={n} should be >= n_clusters={n_clusters}.")
    first, u = sklearn_stream(int(seed), int(n_init), int(n_clusters), n)
    labels, centres, info = ops.kmeans_fit(X, n_clusters, first, u, max_iter=max_iter, tol=tol)
    if return_info:
        return labels, centres, info["inertia"], info
    return labels, centres, info["inertia"]
