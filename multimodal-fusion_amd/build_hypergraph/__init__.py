"""Drop-in mirror of the reference package `build_hypergraph` (build_hypergraph/__init__.py:5-46):
the same public names for the arithmetic path; the HDF5 pipeline functions are SURVEY.md §8(f1)."""
from .similarity_kernel import (build_hypergraph_data, build_weighted_hypergraph, compute_combined_similarity,
                                compute_morphological_similarity, compute_spatial_similarity,
                                mean_pool_with_similarity)
from .preprocess_hypergraph import (aggregate_wsi_super_patches, build_hypergraph_knn_kmeans,
                                    compute_wsi_tma_similarity, group_by_similarity)

__all__ = [
    "compute_morphological_similarity", "compute_spatial_similarity", "compute_combined_similarity",
    "build_weighted_hypergraph", "mean_pool_with_similarity", "build_hypergraph_data",
    "aggregate_wsi_super_patches", "compute_wsi_tma_similarity", "group_by_similarity",
    "build_hypergraph_knn_kmeans",
]
