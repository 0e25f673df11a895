"""Device plumbing shared by the reference-signature mirrors.

The reference resolves `device=None` to "the input's device if it is CUDA, else CPU" and returns its
outputs there (build_hypergraph/preprocess_hypergraph.py:125-126, 235-236, 370-371;
build_hypergraph/similarity_kernel.py:163-168).  The mirrors keep that contract for WHERE RESULTS
LIVE, but the arithmetic always runs through libmmf_hg.so on a ROCm device: a CPU tensor is moved to
the current GPU, computed there and the result moved back.  Without a GPU the call raises
RuntimeError — there is no host implementation to fall back to.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def compute_device(*tensors: torch.Tensor) -> torch.device:
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("multimodal-fusion_amd needs a ROCm GPU: the hypergraph kernels have no CPU path "
                           "(tensors were on CPU and no device is visible)")
    return torch.device("cuda", torch.cuda.current_device())


def result_device_like_kernel(features: torch.Tensor, device: Optional[torch.device]) -> torch.device:
    """similarity_kernel.py:163-164: `device = features.device` when None."""
    return torch.device(device) if device is not None else features.device


def result_device_like_preprocess(t: torch.Tensor, device: Optional[torch.device]) -> torch.device:
    """preprocess_hypergraph.py:125-126: the input's device if CUDA else CPU."""
    if device is not None:
        return torch.device(device)
    return t.device if t.is_cuda else torch.device("cpu")


def to_gpu(t: torch.Tensor, dev: torch.device) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def f32_ceil(x: float) -> float:
    """Smallest float32 >= x.  For a float32 value K and a Python float (double) threshold,
    `K < x` (what the reference's .item() loop evaluates, similarity_kernel.py:195-198) is the same
    as `K < f32_ceil(x)`, so the device comparison stays in float32 without changing one decision."""
    if x != x:
        return x
    t = np.float32(x)
    if float(t) < x:
        t = np.nextafter(t, np.float32(np.inf))
    return float(t)
