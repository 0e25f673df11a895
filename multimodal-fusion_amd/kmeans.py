"""Device KMeans: k-means++ seeding + Lloyd iterations on the GPU.

The reference clusters with ``sklearn.cluster.KMeans(n_clusters, random_state=42, n_init=10)`` on
the host (build_hypergraph/preprocess_hypergraph.py:150-151, 299-300, 391-392).  Here the two
distance computations that dominate it run on the hot-path kernels of libmmf_hg.so:

* assignment  = fused similarity + top-1 (``mmf_simtopk``, squared L2, k = 1) of every point against
  the centroids — labels and distances come out of one scan, the N x k matrix is never stored;
* k-means++    = all n_init seedings in lockstep inside the library (``mmf_kmeanspp_seed``: draw, distance rows,
  choice — three launches per seeding step, no host round trip).

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


def _kmeanspp(X: torch.Tensor, k: int, n_init: int, gen: torch.Generator) -> torch.Tensor:
    """Greedy k-means++ (Arthur & Vassilvitskii, with 2 + log k local trials per step) for ALL n_init seedings in
    lockstep, inside the library (``mmf_kmeanspp_seed``): per step one kernel draws every seeding's trial candidates, one
    forms their clamped distance rows and potentials, one keeps the best trial — the k - 1 steps are enqueued by a single
    call and nothing returns to the host (one seeding at a time with a `.item()` per step was 85 % of a fit; the same
    lockstep in a dozen torch ops per step was still launch-bound: 22 ms at N = 16384).  The uniforms come from the
    caller's generator.  Returns the centres, [n_init, k, D]."""
    trials = 2 + int(math.log(k))
    u_first = torch.rand(n_init, generator=gen, device=X.device)
    u_steps = torch.rand((k - 1, n_init, trials), generator=gen, device=X.device)
    return X[ops.kmeanspp_seed(X, k, u_first, u_steps)].contiguous()


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
    C = _kmeanspp(Xc, n_clusters, n_init, gen)                                # [n_init, k, D]
    # lockstep while the [n_init * k, N] distance rows stay within 512 MiB (measured with the tiled distance kernel:
    # N = 65536, k = 100: 61 -> 40 ms per fit; 10 groups of 100 rows: 8 -> 2 ms); beyond that restart by restart
    if n_init * n_clusters <= 16384 and n_init * n_clusters * n <= (1 << 27):
        labels, C, inertia = _lloyd_lockstep(Xc, C, max_iter, tol_abs)
    else:
        labels, C, inertia = _lloyd_one_by_one(Xc, C, max_iter, tol_abs)
    return labels, C + mean, inertia


def _lloyd_lockstep(Xc: torch.Tensor, C: torch.Tensor, max_iter: int, tol_abs: float):
    """All restarts iterate together: one kernel forms the distance rows of every restart's centroids
    (``mmf_seed_distances`` with explicit rows), one counting sort groups the members of all n_init * k clusters, one
    segmented mean updates all centroids — a dozen launches and two host round trips per iteration for ALL restarts
    (one restart at a time: that many per restart).  A restart that has converged keeps iterating at its fixed point
    until the last one has."""
    I, k, d = C.shape
    n = Xc.shape[0]
    base = (torch.arange(I, device=Xc.device) * k)[:, None]
    def assign(C):
        D = ops.seed_distances(Xc, C.reshape(I * k, d)).view(I, k, n)
        d2, lab = D.min(dim=1)                                                # [I, n]
        return lab, d2
    for _it in range(max_iter):
        lab, d2 = assign(C)
        seg = ops.segment_sort((lab + base).reshape(-1), I * k)               # members of all I * k clusters
        rows = ops.Segments(seg.counts, seg.offsets, seg.order % n, n, I * k)
        newC = ops.segment_mean(Xc, rows).view(I, k, d)
        empty = (seg.counts == 0).view(I, k)
        shift = ((newC - C) ** 2).sum(dim=(1, 2))
        any_empty, done = torch.stack([empty.any(), (shift <= tol_abs).all()]).tolist()
        if any_empty:                                                         # relocate empty clusters to the points
            for i in torch.nonzero(empty.any(dim=1)).flatten().tolist():      # farthest from their centre
                far = torch.topk(d2[i], int(empty[i].sum())).indices
                newC[i][empty[i]] = Xc[far]
            done = False
        C = newC
        if done:
            break
    lab, d2 = assign(C)
    inertia = d2.sum(dim=1)
    best = int(torch.argmin(inertia))
    return lab[best].contiguous(), C[best], float(inertia[best])


def _lloyd_one_by_one(Xc: torch.Tensor, seeds: torch.Tensor, max_iter: int, tol_abs: float):
    """Large problems: one restart at a time, assignment by the fused similarity + top-1 scan, which never stores the
    N x k matrix (and stops each restart at its own convergence)."""
    n_clusters = seeds.shape[1]
    best = None
    for init in range(seeds.shape[0]):
        C = seeds[init]
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
            best = (labels, C, inertia)
    return best
