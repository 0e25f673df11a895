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


def _gather_into(out: torch.Tensor, inp: torch.Tensor, group, async_op: bool = False):
    """all_gather_into_tensor, with host staging under gloo (rehearsal only)."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=group)
        out.copy_(o)
        return None
    return dist.all_gather_into_tensor(out, inp.contiguous(), group=group, async_op=async_op)


def _allreduce_max(t: torch.Tensor, group) -> None:
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


def _overlapped_simtopk(x_local, n_total, lo, hi, world, *, metric, lam, k, exclude_self, operand, group, return_stats):
    """Phase path: the f32 all-gather — needed only by the exact re-rank — is started first
    and left in flight; each rank prepares the 16-bit operands of its own rows, ranks exchange those
    (half the bytes) plus five floats per row, and the scan runs on them.  The stream waits for the f32
    rows only between the scan and the re-rank."""
    from . import ops
    dev = x_local.device
    rows, d = x_local.shape
    dp = ops.padded_dim(d)
    # local phase
    scal_l = torch.empty((rows,), dtype=torch.float32, device=dev)
    maxn = torch.zeros((1,), dtype=torch.float32, device=dev)
    ops.row_scalars(x_local, metric, scal_l, maxn)
    if metric != "cosine":
        _allreduce_max(maxn, group)                      # the common power-of-two scale needs the global maximum
    z16 = torch.float16 if operand == "f16" else torch.bfloat16
    z_l = torch.empty((rows, dp), dtype=z16, device=dev)
    pack_l = torch.empty((5, rows), dtype=torch.float32, device=dev)      # scal, zn, rn, un, cb
    pack_l[0] = scal_l
    max4 = torch.zeros((4,), dtype=torch.float32, device=dev)
    ops.prep_rows(x_local, metric, operand, scal_l, maxn, z_l, pack_l[1], pack_l[2], pack_l[3], pack_l[4], max4)
    # exchange of the prepared operands
    m_pad = (n_total + 255) // 256 * 256
    z_full = torch.zeros((m_pad + 256, dp), dtype=z16, device=dev)
    _gather_into(z_full[:n_total], z_l, group)
    pack_full = torch.empty((world * 5, rows), dtype=torch.float32, device=dev)
    _gather_into(pack_full, pack_l, group)
    pf = pack_full.view(world, 5, rows).permute(1, 0, 2).reshape(5, n_total)
    side = {}
    for i, name in enumerate(("scal", "zn", "rn", "un", "cb")):
        buf = torch.full((m_pad + 256,), float("-inf") if name == "cb" else 0.0, dtype=torch.float32, device=dev)
        buf[:n_total] = pf[i]
        side[name] = buf
    _allreduce_max(max4, group)
    # Only now the big f32 gather: collectives of one group run in issue order on RCCL's stream, so it
    # must not sit in front of the small exchanges the scan is waiting for.  It then overlaps the scan.
    full = torch.empty((n_total, d), dtype=x_local.dtype, device=dev)
    w_full = _gather_into(full, x_local, group, async_op=True)
    c = dict(Z=z_full, **side)
    q = dict(Z=z_full[lo:], scal=side["scal"][lo:], zn=side["zn"][lo:], rn=side["rn"][lo:], un=side["un"][lo:],
             cb=side["cb"][lo:])
    ev = None
    if w_full is not None:                                # the f32 rows: wait only after the scan
        ev = torch.cuda.Event()
        side_stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side_stream):
            w_full.wait()
            ev.record(side_stream)
    out = ops.simtopk_prepared(full[lo:hi], full, q, c, m_pad, max4, operand=operand, metric=metric, lam=lam, k=k,
                               exclude_self=exclude_self, row_offset=lo, col_offset=0, wait_event=ev,
                               profile=return_stats, return_stats=return_stats)
    return out


def sharded_simtopk(x_local: torch.Tensor, n_total: int, *, metric="cosine", lam: float = 1.0, k: int = 5,
                    exclude_self: bool = True, precision: str = "auto", group=None, gather_output: bool = False,
                    op: Optional[Callable] = None, return_stats: bool = False, overlap: Optional[bool] = None):
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
    if op is None and world > 1 and overlap is not False and x_local.is_cuda and isinstance(metric, str):
        from . import ops as _ops
        kk = k + (1 if exclude_self else 0)
        equal = (n_total % world) == 0
        if equal and precision in ("auto", "fast", "fast_bf16") and _ops.padded_dim(x_local.shape[1]) > 0 and kk <= 8 \
                and x_local.dtype == torch.float32:
            out = _overlapped_simtopk(x_local.contiguous(), n_total, lo, hi, world, metric=metric, lam=lam, k=k,
                                      exclude_self=exclude_self, operand="bf16" if precision == "fast_bf16" else "f16",
                                      group=group, return_stats=return_stats)
            idx, val = out[0], out[1]
            if gather_output:
                idx = all_gather_rows(idx, n_total, group)
                val = all_gather_rows(val, n_total, group)
            return (idx, val, out[2]) if return_stats else (idx, val)
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
