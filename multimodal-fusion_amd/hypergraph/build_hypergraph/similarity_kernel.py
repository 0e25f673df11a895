"""hypergraph/build_hypergraph/similarity_kernel.py of the reference, signature variants only."""
from __future__ import annotations

from typing import Optional

import torch

from ...build_hypergraph.similarity_kernel import (build_weighted_hypergraph, compute_combined_similarity,  # noqa: F401
                                                   compute_morphological_similarity, compute_spatial_similarity)


def mean_pool_with_similarity(features: torch.Tensor, positions: torch.Tensor, lambda_h: float = 1.0,
                              lambda_g: float = 1.0) -> torch.Tensor:
    """4-argument variant (:214-247); positions and lambdas are unused there as well."""
    return torch.mean(features, dim=0, keepdim=True)


def build_hypergraph_data(features: torch.Tensor, positions: torch.Tensor, lambda_h: float = 1.0,
                          lambda_g: float = 1.0, threshold_median_ratio: float = None, use_pooling: bool = True,
                          device: Optional[torch.device] = None) -> dict:
    if device is None:
        device = features.device
    features = features.to(device)
    positions = positions.to(device)
    edge_index, edge_weights = build_weighted_hypergraph(features, positions, lambda_h, lambda_g,
                                                         threshold_median_ratio, device)
    result = {"x": features, "edge_index": edge_index, "edge_attr": edge_weights, "pos": positions}
    if use_pooling:
        result["pooled_features"] = mean_pool_with_similarity(features, positions, lambda_h, lambda_g)
    return result
