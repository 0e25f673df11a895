"""Device KMeans: k-means++ seeding + Lloyd iterations on the GPU.

The reference clusters with ``sklearn.cluster.KMeans(n_clusters, random_state=42, n_init=10)`` on
the host (build_hypergraph/preprocess_hypergraph.py:150-151, 299-300, 391-392).  Here the two
distance computations that dominate it run on the hot-path kernels of libmmf_hg.so:

* assignment  = fused similarity + top-1 (``mmf_simtopk``, squared L2, k = 1) of every point against
  the centroids — labels and distances come out of one scan, the N x k matrix is never stored;
* k-means++    = dense squared-L2 rows of the few candidate centres against all points (``mmf_sim_dense``).

Centroid updates are segmented means in a fixed summation order (``mmf_segment_sort`` + ``mmf_segment_mean``), so the
whole fit is deterministic: same data and seed, same labels, run after run.  Bit-parity with scikit-learn is not
achievable (its seeding consumes a Mersenne-Twister stream); parity is on the objective: same
partition on separable data, inertia within a few per cent otherwise (tests/test_gpu_kmeans.py).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from . import ops


def _assign(X: torch.Tensor, C: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    idx, val = ops.simtopk(X, C, metric="neg_sq_l2", k=1, exclude_self=False)
    return idx[:, 0], (-val[:, 0]).clamp_min_(0.0)


def _sq_dists(A: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    return (-ops.sim_dense(A, X, metric="neg_sq_l2")).clamp_min_(0.0)        # [len(A), N]


def _kmeanspp(X: torch.Tensor, k: int, gen: torch.Generator) -> torch.Tensor:
    """Greedy k-means++ (Arthur & Vassilvitskii, with 2 + log k local trials per step)."""
    n = X.shape[0]
    trials = 2 + int(math.log(k))
    first = int(torch.randint(n, (1,), generator=gen, device=X.device))
    centers = [X[first]]
    closest = _sq_dists(X[first:first + 1], X)[0]
    pot = closest.sum()
    for _ in range(1, k):
        r = torch.rand(trials, generator=gen, device=X.device) * pot
        cand = torch.searchsorted(torch.cumsum(closest, 0), r).clamp_(max=n - 1)
        dc = torch.minimum(_sq_dists(X[cand], X), closest[None, :])        # [trials, N]
        pots = dc.sum(dim=1)
        best = int(torch.argmin(pots))
        closest, pot = dc[best], pots[best]
        centers.append(X[cand[best]])
    return torch.stack(centers, 0).contiguous()


def kmeans_fit_predict(X: torch.Tensor, n_clusters: int, *, n_init: int = 10, max_iter: int = 300,
                       tol: float = 1e-4, seed: int = 42) -> Tuple[torch.Tensor, torch.Tensor, float]:
    """Returns (labels int64 [N], centers f32 [k, D], inertia).  X must live on a ROCm device."""
    if not X.is_cuda:
        raise RuntimeError("kmeans_fit_predict: X must be on a ROCm device (no CPU path)")
    X = X.detach().float().contiguous()
    n, d = X.shape
    if not (1 <= n_clusters <= n):
        raise ValueError(f"n_samples={n} should be >= n_clusters={n_clusters}.")
    mean = X.mean(dim=0, keepdim=True)
    Xc = X - mean                                                             # as sklearn: better conditioned
    tol_abs = float(tol) * float(Xc.var(dim=0, unbiased=False).mean())
    gen = torch.Generator(device=X.device).manual_seed(int(seed))
    best = None
    for _ in range(n_init):
        C = _kmeanspp(Xc, n_clusters, gen)
        labels = None
        for _it in range(max_iter):
            labels, d2 = _assign(Xc, C)
            seg = ops.segment_sort(labels, n_clusters)
            newC = ops.segment_mean(Xc, seg)
            empty = seg.counts == 0
            if bool(empty.any()):                                            # relocate empty clusters to the
                far = torch.topk(d2, int(empty.sum())).indices               # points farthest from their centre
                newC[empty] = Xc[far]
            shift = float(((newC - C) ** 2).sum())
            C = newC
            if shift <= tol_abs:
                break
        labels, d2 = _assign(Xc, C)
        inertia = float(d2.sum())
        if best is None or inertia < best[2]:
            best = (labels, C + mean, inertia)
    return best
