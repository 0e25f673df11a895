"""Row-sharded multi-GPU similarity + top-k: one process per GPU, rank r owns the rows
[r*N/P, (r+1)*N/P) and their [N/P, k] outputs.  The only exchange is ONE all-gather of the feature
shard (RCCL over xGMI when the backend is "nccl"); each rank then scans every column for its own
rows with mmf_simtopk and the global row/column offsets.  No all-reduce, no all-to-all: outputs are
row-owned and stay sharded unless `gather_output=True`.

The reference has no distributed code; correctness here means
sharded(P) == unsharded, bit for bit, which tests/test_distributed_cpu.py checks with gloo and the
oracle standing in for the device op, and tests/test_gpu_parity.py checks through the offsets.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row blocks; the first n_total % world ranks get one extra row."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(x_local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather row shards of possibly unequal height into the full [n_total, d] matrix."""
    world = dist.get_world_size(group)
    if world == 1:
        return x_local
    d = x_local.shape[1]
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    heights = [hi - lo for lo, hi in sizes]
    if x_local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (gloo has no device collectives): stage through the host.  The production
        # backend is "nccl" (= RCCL over xGMI on ROCm) and never takes this branch.
        return all_gather_rows(x_local.cpu(), n_total, group).to(x_local.device)
    full = torch.empty((n_total, d), dtype=x_local.dtype, device=x_local.device)
    if len(set(heights)) == 1:
        dist.all_gather_into_tensor(full, x_local.contiguous(), group=group)
    else:
        outs = [full[lo:hi] for lo, hi in sizes]
        # all_gather wants equal shapes: pad to the tallest shard, then copy the valid part
        hmax = max(heights)
        pad = torch.zeros((hmax, d), dtype=x_local.dtype, device=x_local.device)
        pad[: x_local.shape[0]] = x_local
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        for o, b, h in zip(outs, bufs, heights):
            o.copy_(b[:h])
    return full


def sharded_simtopk(x_local: torch.Tensor, n_total: int, *, metric="cosine", lam: float = 1.0, k: int = 5,
                    exclude_self: bool = True, precision: str = "auto", group=None, gather_output: bool = False,
                    op: Optional[Callable] = None, return_stats: bool = False):
    """Top-k of every local row against ALL n_total rows.

    x_local: this rank's [N_r, d] shard (rows shard_bounds(n_total, world, rank)).
    Returns (idx int64 [N_r, k] GLOBAL column ids, val f32 [N_r, k]); with gather_output the full
    [n_total, k] result is replicated on every rank.
    `op` lets the CPU tests substitute the oracle for the device op; the default is the HIP path.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_total, world, rank)
    if x_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: shard has {x_local.shape[0]} rows, expected {hi - lo}")
    full = all_gather_rows(x_local, n_total, group) if world > 1 else x_local
    if op is None:
        from . import ops
        op = ops.simtopk
    # hand the op the local rows as a VIEW of the gathered matrix: libmmf_hg.so recognises a row
    # slice of Y and prepares the operands once
    x_rows = full[lo:hi] if world > 1 else x_local
    kw = dict(metric=metric, lam=lam, k=k, exclude_self=exclude_self, row_offset=lo, col_offset=0)
    stats = None
    if op.__module__.endswith("ops"):
        out = op(x_rows, full, precision=precision, return_stats=return_stats, profile=return_stats, **kw)
        if return_stats:
            idx, val, stats = out
        else:
            idx, val = out
    else:
        idx, val = op(x_rows, full, **kw)
    if gather_output and world > 1:
        idx = all_gather_rows(idx, n_total, group)
        val = all_gather_rows(val, n_total, group)
    if return_stats:
        return idx, val, stats
    return idx, val
