"""Drop-in mirror of the reference package `build_hypergraph` (build_hypergraph/__init__.py:5-46): the same 17 public
names, same signatures (tests/golden/signatures.json), arithmetic on the gfx950 kernels of libmmf_hg.so."""
from .similarity_kernel import (build_hypergraph_data, build_weighted_hypergraph, compute_combined_similarity,
                                compute_morphological_similarity, compute_spatial_similarity,
                                mean_pool_with_similarity)
from .preprocess_hypergraph import (aggregate_wsi_super_patches, batch_rebuild_hypergraph, build_hypergraph_knn_kmeans,
                                    compute_wsi_tma_similarity, group_by_similarity, load_similarity_matrices,
                                    load_tma_data, load_wsi_data, process_dataset, process_single_file,
                                    rebuild_hypergraph_from_similarity, save_hypergraph_to_h5, set_kmeans_backend)
from . import h5io  # noqa: F401

__all__ = [
    "compute_morphological_similarity",
    "compute_spatial_similarity",
    "compute_combined_similarity",
    "build_weighted_hypergraph",
    "mean_pool_with_similarity",
    "process_single_file",
    "process_dataset",
    "load_wsi_data",
    "load_tma_data",
    "aggregate_wsi_super_patches",
    "compute_wsi_tma_similarity",
    "group_by_similarity",
    "build_hypergraph_knn_kmeans",
    "save_hypergraph_to_h5",
    "load_similarity_matrices",
    "rebuild_hypergraph_from_similarity",
    "batch_rebuild_hypergraph",
]
