"""Mirror of the reference's build_hypergraph/similarity_kernel.py — same names, argument order,
defaults and error behaviour; the arithmetic runs in the gfx950 kernels behind include/mmf_hg.h.

    reference                                   here
    compute_morphological_similarity  :17-54    mmf_sim_dense(MMF_RBF)         (f32 MFMA, canonical chain)
    compute_spatial_similarity        :57-86    mmf_sim_dense(MMF_RBF) on the 2-/3-D positions
    compute_combined_similarity       :88-124   mmf_sim_dense_combined         (one pass, no K_h/K_g temporaries)
    build_weighted_hypergraph         :126-212  mmf_offdiag_lower_median + mmf_threshold_edges (materialised K), or
                                                mmf_combined_offdiag_median + mmf_combined_threshold_edges (large N)
    mean_pool_with_similarity         :214-238  torch.mean (trivial, a5)
    build_hypergraph_data             :240-306  packaging
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import ops
from ._common import compute_device, f32_ceil, result_device_like_kernel, to_gpu

# build_weighted_hypergraph keeps K = K_h * K_g in HBM up to this many bytes; beyond it the median and the edges are
# computed from K recomputed in row panels of PANEL_ROWS rows (0 = about 1 GiB per panel).  Same result either way.
STREAM_BYTES = 32 << 30
PANEL_ROWS = 0


def compute_morphological_similarity(features: torch.Tensor, lambda_h: float = 1.0) -> torch.Tensor:
    """K_h[i,j] = exp(-lambda_h * ((n_i + n_j) - 2 h_i.h_j)), [N, N] f32 (similarity_kernel.py:43-52)."""
    dev = compute_device(features)
    K = ops.sim_dense(to_gpu(features, dev), metric="rbf", lam=float(lambda_h))
    return K.to(features.device)


def compute_spatial_similarity(positions: torch.Tensor, lambda_g: float = 1.0) -> torch.Tensor:
    """K_g from [N, 2] or [N, 3] positions (similarity_kernel.py:79-84)."""
    dev = compute_device(positions)
    K = ops.sim_dense(to_gpu(positions, dev), metric="rbf", lam=float(lambda_g))
    return K.to(positions.device)


def compute_combined_similarity(features: torch.Tensor, positions: torch.Tensor, lambda_h: float = 1.0,
                                lambda_g: float = 1.0) -> torch.Tensor:
    """K = K_h * K_g (similarity_kernel.py:116-122), fused into one kernel."""
    dev = compute_device(features, positions)
    K = ops.sim_dense_combined(to_gpu(features, dev), to_gpu(positions, dev), float(lambda_h), float(lambda_g))
    return K.to(features.device)


def build_weighted_hypergraph(features: torch.Tensor, positions: torch.Tensor, lambda_h: float = 1.0,
                              lambda_g: float = 1.0, threshold_median_ratio: float = None,
                              device: Optional[torch.device] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Edges (i, j) with K[i, j] >= median_offdiag(K) * ratio, row-major, self loops kept
    (similarity_kernel.py:171-212).  As in the reference: N <= 1 raises ValueError (:176-178) and the
    default ratio None raises TypeError at `median * None` (:188)."""
    out_dev = result_device_like_kernel(features, device)
    dev = compute_device(features, positions) if out_dev.type != "cuda" else out_dev
    N = features.shape[0]
    if N * N * 4 > STREAM_BYTES:
        # K would not be worth (or possible) keeping: recompute it in row panels — same median, same edges
        F, Pz = to_gpu(features, dev), to_gpu(positions, dev)
        if N <= 1:
            raise ValueError(f"Number of nodes must be greater than 1, got N={N}. "
                             f"Hypergraph construction requires at least 2 nodes.")
        median_sim = ops.combined_offdiag_median(F, Pz, float(lambda_h), float(lambda_g), PANEL_ROWS).item()
        threshold = median_sim * threshold_median_ratio
        edge_index, edge_weights = ops.combined_threshold_edges(F, Pz, f32_ceil(threshold), float(lambda_h), float(lambda_g),
                                                                PANEL_ROWS)
        return edge_index.to(out_dev).contiguous(), edge_weights.to(out_dev)
    K = ops.sim_dense_combined(to_gpu(features, dev), to_gpu(positions, dev), float(lambda_h), float(lambda_g))
    N = K.shape[0]
    if N <= 1:
        raise ValueError(f"Number of nodes must be greater than 1, got N={N}. "
                         f"Hypergraph construction requires at least 2 nodes.")
    median_sim = ops.offdiag_lower_median(K).item()
    threshold = median_sim * threshold_median_ratio            # TypeError when the ratio is None, as upstream
    edge_index, edge_weights = ops.threshold_edges(K, f32_ceil(threshold))
    return edge_index.to(out_dev).contiguous(), edge_weights.to(out_dev)


def mean_pool_with_similarity(features: torch.Tensor) -> torch.Tensor:
    """Global mean feature, [1, D] (similarity_kernel.py:236)."""
    return torch.mean(features, dim=0, keepdim=True)


def build_hypergraph_data(features: torch.Tensor, positions: torch.Tensor, lambda_h: float = 1.0,
                          lambda_g: float = 1.0, threshold_median_ratio: float = None, use_pooling: bool = True,
                          device: Optional[torch.device] = None) -> dict:
    """similarity_kernel.py:283-306; the pooled entry is keyed 'pooled_feature' in this copy."""
    if device is None:
        device = features.device
    features = features.to(device)
    positions = positions.to(device)
    edge_index, edge_weights = build_weighted_hypergraph(features, positions, lambda_h, lambda_g,
                                                         threshold_median_ratio, device)
    result = {"x": features, "edge_index": edge_index, "edge_attr": edge_weights, "pos": positions}
    if use_pooling:
        result["pooled_feature"] = mean_pool_with_similarity(features)
    return result
