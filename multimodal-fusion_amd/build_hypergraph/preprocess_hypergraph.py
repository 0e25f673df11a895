"""Mirror of the arithmetic functions of the reference's build_hypergraph/preprocess_hypergraph.py.

    reference                                  here
    compute_wsi_tma_similarity     :202-267    mmf_sim_dense(MMF_RBF_DIRECT) + device reductions for the stats
    build_hypergraph_knn_kmeans    :335-433    mmf_simtopk(MMF_NEG_SQ_L2) for the k-NN (:379-388),
                                               device dedup (:403-404), mmf_edge_cosine for the weights (:414-420)
    group_by_similarity            :270-332    } KMeans steps: SURVEY.md §8(f3) "next"; they keep the
    aggregate_wsi_super_patches    :87-199     } reference's own scikit-learn call on the host for now,
                                                 everything around it (similarity, pooling, stats) is on device

Documented divergences: self is dropped from the k-NN by identity instead of
"column 0" (A5); edges come out lexicographically sorted instead of in Python set order (A6);
stats hold Python ints (A3).  The HDF5 functions of the file are not mirrored yet (§8 f1).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import ops
from ._common import compute_device, result_device_like_preprocess, to_gpu
from .similarity_kernel import compute_combined_similarity


KMEANS_BACKEND = "sklearn"     # "sklearn": the reference's own call on the host (labels identical to the reference)
                               # "device" : multimodal-fusion_amd/kmeans.py (same objective, different seeding stream)


def set_kmeans_backend(name: str) -> None:
    global KMEANS_BACKEND
    if name not in ("sklearn", "device"):
        raise ValueError("kmeans backend must be 'sklearn' or 'device'")
    KMEANS_BACKEND = name


def _kmeans_labels(x, n_clusters: int) -> np.ndarray:
    if KMEANS_BACKEND == "device":
        from ..kmeans import kmeans_fit_predict
        xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))
        if not xt.is_cuda:
            xt = xt.to(compute_device())
        return kmeans_fit_predict(xt, n_clusters, n_init=10, seed=42)[0].cpu().numpy()
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    # the reference's exact call (preprocess_hypergraph.py:150-151, 299-300, 391-392)
    try:
        from sklearn.cluster import KMeans
    except ImportError as e:  # pragma: no cover
        raise ImportError("KMeans steps still use scikit-learn on the host (SURVEY.md §8 f3)") from e
    return KMeans(n_clusters=n_clusters, random_state=42, n_init=10).fit_predict(x)


def _matrix_stats(S: torch.Tensor) -> Dict[str, float]:
    return {"mean": S.mean().item(), "std": S.std().item(), "min": S.min().item(), "max": S.max().item(),
            "median": S.median().item()}


def compute_wsi_tma_similarity(wsi_features: torch.Tensor, wsi_positions: torch.Tensor, tma_features: torch.Tensor,
                               lambda_h: float = 1.0, lambda_g: float = 1.0,
                               device: Optional[torch.device] = None) -> Tuple[torch.Tensor, Dict]:
    """[N_wsi, N_tma] exp(-lambda_h * sum_k (a_k - b_k)^2) + its statistics (:248-265).
    `wsi_positions` and `lambda_g` are accepted and ignored, as in the reference (Appendix A7)."""
    out_dev = result_device_like_preprocess(wsi_features, device)
    dev = out_dev if out_dev.type == "cuda" else compute_device(wsi_features, tma_features)
    S = ops.sim_dense(to_gpu(wsi_features, dev), to_gpu(tma_features, dev), metric="rbf_direct", lam=float(lambda_h))
    return S.to(out_dev), _matrix_stats(S)


def group_by_similarity(similarity_matrix: torch.Tensor, num_groups: int, method: str = "kmeans"):
    """KMeans over the rows of the similarity matrix (:297-306).  method='knn' is the reference's
    broken branch (Appendix A2) and is not provided."""
    if method != "kmeans":
        raise ValueError(f"Unknown grouping method: {method}")
    labels = _kmeans_labels(similarity_matrix.detach().cpu().numpy(), num_groups)
    stats = {"method": "kmeans", "num_groups": num_groups,
             "group_sizes": [int(np.sum(labels == i)) for i in range(num_groups)]}
    return labels, stats


def aggregate_wsi_super_patches(wsi_features: torch.Tensor, wsi_positions: torch.Tensor, num_super_patches: int,
                                lambda_h: float = 1.0, lambda_g: float = 1.0, device: Optional[torch.device] = None,
                                wsi_similarity_matrix: Optional[torch.Tensor] = None):
    """Cluster patches (KMeans on the features, :150-151), mean-pool each cluster (:157-170) and report
    intra-cluster / whole-matrix similarity statistics (:172-197).  Returns
    (super_features, super_positions, stats, K_wsi) on the reference's result device."""
    out_dev = result_device_like_preprocess(wsi_features, device)
    dev = out_dev if out_dev.type == "cuda" else compute_device(wsi_features, wsi_positions)
    F = to_gpu(wsi_features, dev)
    P = to_gpu(wsi_positions, dev)
    N = F.shape[0]
    K = to_gpu(wsi_similarity_matrix, dev) if wsi_similarity_matrix is not None else \
        ops.sim_dense_combined(F, P, float(lambda_h), float(lambda_g))
    labels = torch.from_numpy(_kmeans_labels(F.cpu().numpy(), num_super_patches)).to(dev)
    counts = torch.bincount(labels, minlength=num_super_patches)
    if int(counts.min()) == 0:
        raise ValueError(f"Cluster {int(torch.argmin(counts))} is empty")
    onehot = torch.zeros((num_super_patches, N), dtype=torch.float32, device=dev)
    onehot[labels, torch.arange(N, device=dev)] = 1.0
    inv = 1.0 / counts.to(torch.float32)[:, None]
    super_f = (onehot @ F) * inv
    super_p = (onehot @ P) * inv
    # mean off-diagonal similarity inside every cluster with more than one member (:175-184)
    intra = []
    block_sum = onehot @ K @ onehot.t()
    diag_sum = onehot @ torch.diagonal(K)
    for c in range(num_super_patches):
        m = int(counts[c])
        if m > 1:
            intra.append(((block_sum[c, c] - diag_sum[c]) / (m * (m - 1))).item())
    stats = {"num_original_patches": int(N), "num_super_patches": int(num_super_patches),
             "avg_intra_cluster_similarity": float(np.mean(intra)) if intra else 0.0,
             "wsi_similarity_matrix_stats": _matrix_stats(K)}
    return super_f.to(out_dev), super_p.to(out_dev), stats, K.to(out_dev)


def build_hypergraph_knn_kmeans(wsi_features: torch.Tensor, tma_features: torch.Tensor, group_labels: np.ndarray,
                                k: int = 5, num_hyperedges: int = 10,
                                device: Optional[torch.device] = None) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """k-NN edges + KMeans cliques, undirected dedup, max(0, cosine) weights (:373-433).
    `group_labels` is accepted and ignored, as in the reference (Appendix A7)."""
    out_dev = result_device_like_preprocess(wsi_features, device)
    dev = out_dev if out_dev.type == "cuda" else compute_device(wsi_features, tma_features)
    all_f = torch.cat([to_gpu(wsi_features, dev), to_gpu(tma_features, dev)], dim=0)
    n_total, n_wsi = all_f.shape[0], wsi_features.shape[0]
    if k + 1 > n_total:   # what sklearn's kneighbors raises at :382
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k + 1}, "
                         f"n_samples_fit = {n_total}, n_samples = {n_total}")
    # (a8) Euclidean k-NN, self dropped by identity
    nbr, _ = ops.simtopk(all_f, metric="neg_sq_l2", k=k, exclude_self=True)
    src = torch.arange(n_total, device=dev, dtype=torch.int64)[:, None].expand(n_total, k)
    pairs = [torch.stack([src.reshape(-1), nbr.reshape(-1)], dim=0)]
    # (a10) cliques of the KMeans hyperedges; the clustering itself is the reference's sklearn call
    labels = torch.from_numpy(_kmeans_labels(all_f.cpu().numpy(), num_hyperedges)).to(dev)
    for h in range(num_hyperedges):
        nodes = torch.nonzero(labels == h, as_tuple=False).reshape(-1)
        if nodes.numel() > 1:
            comb = torch.combinations(nodes, r=2)
            pairs.append(comb.t())
    e = torch.cat(pairs, dim=1)
    # (a9) undirected dedup: tuple(sorted(edge)) through a set (:403-404) == unique of (min, max)
    lo, hi = torch.minimum(e[0], e[1]), torch.maximum(e[0], e[1])
    code = torch.unique(lo * n_total + hi)
    edge_index = torch.stack([code // n_total, code % n_total], dim=0).contiguous()
    if edge_index.shape[1] == 0:
        edge_index = torch.empty((2, 0), dtype=torch.long, device=dev)
        edge_weights = torch.empty((0,), dtype=torch.float32, device=dev)
    else:
        edge_weights = ops.edge_cosine(all_f, edge_index)
    stats = {"num_nodes": int(n_total), "num_wsi_super_patches": int(n_wsi),
             "num_tma_patches": int(tma_features.shape[0]), "num_edges": int(edge_index.shape[1]),
             "num_hyperedges": int(num_hyperedges), "k": int(k)}
    return edge_index.to(out_dev), edge_weights.to(out_dev), stats
