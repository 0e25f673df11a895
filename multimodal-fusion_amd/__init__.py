"""multimodal-fusion_amd — MI355X-native hypergraph construction (similarity + per-row top-k).

The directory name carries a hyphen (it mirrors the reference repo's name), so import it through the
shim at the repo root:  ``import multimodal_fusion_amd as mmf``.

    mmf.ops.simtopk(...)                          fused similarity + top-k on gfx950
    mmf.build_hypergraph.*                        the reference's function names and signatures
    mmf.distributed.sharded_simtopk(...)          row-sharded multi-GPU driver (RCCL all-gather)
"""
from . import _lib, ops  # noqa: F401
from .ops import (edge_cosine, offdiag_lower_median, sim_dense, sim_dense_combined, simtopk,  # noqa: F401
                  threshold_edges, topk_merge)

__all__ = ["ops", "simtopk", "sim_dense", "sim_dense_combined", "edge_cosine", "topk_merge",
           "offdiag_lower_median", "threshold_edges"]
