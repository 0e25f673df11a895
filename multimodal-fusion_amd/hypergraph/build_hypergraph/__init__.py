"""Mirror of the reference's SECOND copy, hypergraph/build_hypergraph (__init__.py:5-19).  Same
kernels; the two signature differences of that copy are kept:
mean_pool_with_similarity takes (features, positions, lambda_h, lambda_g) (:214-219) and
build_hypergraph_data stores the pooled row under 'pooled_features' (:315)."""
from ...build_hypergraph.similarity_kernel import (build_weighted_hypergraph, compute_combined_similarity,
                                                   compute_morphological_similarity, compute_spatial_similarity)
from . import similarity_kernel
from .similarity_kernel import build_hypergraph_data, mean_pool_with_similarity

__all__ = ["compute_morphological_similarity", "compute_spatial_similarity", "compute_combined_similarity",
           "build_weighted_hypergraph", "mean_pool_with_similarity"]
