"""Row-sharded multi-GPU similarity + top-k: one process per GPU, rank r owns the rows
[r*N/P, (r+1)*N/P) and their [N/P, k] outputs.  The only exchange is an all-gather of the row shards
(RCCL over xGMI when the backend is "nccl"); each rank then scans every column for its own rows with the
global row/column offsets.  No all-to-all: outputs are row-owned and stay sharded unless `gather_output=True`.

Two drivers: the simple one (one all-gather of the feature shard, then mmf_simtopk) and the pipelined one used
for equal shards (16-bit operands exchanged in chunks under the scan, the feature rows under all of it).

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


def _staged(t: torch.Tensor, group) -> bool:
    """gloo has no device collectives: the rehearsal path stages device tensors through the host.  The
    production backend is "nccl" (= RCCL over xGMI on ROCm) and never stages."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _gather_into(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    """Blocking all_gather_into_tensor on flat views (rank-major concatenation; backends differ in which
    shaped outputs they accept)."""
    if not out.is_contiguous():
        raise ValueError("all-gather output must be contiguous")
    flat_in = inp.contiguous().view(-1)
    if flat_in.numel() * dist.get_world_size(group) != out.numel():
        raise ValueError("all-gather output must hold world_size inputs")
    if _staged(inp, group):
        o = torch.empty((out.numel(),), dtype=out.dtype)
        dist.all_gather_into_tensor(o, flat_in.cpu(), group=group)
        out.view(-1).copy_(o)
        return
    dist.all_gather_into_tensor(out.view(-1), flat_in, group=group)


def _gather_event(out: torch.Tensor, inp: torch.Tensor, group, side_stream) -> "torch.cuda.Event":
    """Issue one all-gather of the pipelined exchange and return the event that fires when `out` is complete.
    nccl: the collective is asynchronous (it runs on the backend's stream behind the work the current stream has
    queued so far); the side stream waits for it and records the event, the current stream is never blocked.
    gloo rehearsal: the gather itself is a blocking host collective; its result is copied to the device ON THE SIDE
    STREAM and the event recorded behind the copy, so consumers go through the same event hand-off
    (mmf_panel.ready_event / select_wait_event) as with nccl."""
    if not out.is_contiguous():
        raise ValueError("all-gather output must be contiguous")
    flat_in = inp.contiguous().view(-1)
    if flat_in.numel() * dist.get_world_size(group) != out.numel():
        raise ValueError("all-gather output must hold world_size inputs")
    ev = torch.cuda.Event()
    if _staged(inp, group):
        host = torch.empty((out.numel(),), dtype=out.dtype)
        dist.all_gather_into_tensor(host, flat_in.cpu(), group=group)       # .cpu() waits for the producer stream
        with torch.cuda.stream(side_stream):
            out.view(-1).copy_(host)
            ev.record(side_stream)
        return ev
    work = dist.all_gather_into_tensor(out.view(-1), flat_in, group=group, async_op=True)
    with torch.cuda.stream(side_stream):
        work.wait()
        ev.record(side_stream)
    return ev


def _allreduce_max(t: torch.Tensor, group) -> None:
    if _staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


# Exchange buffers of the pipelined driver, kept between calls: allocating half a gigabyte per step costs most of a
# millisecond of host time in the caching allocator, which at 8 GPUs is a tenth of the step.  A call ends with the
# library's stream synchronisation, so the previous call's work on them is finished before the next one starts.
# The buffers are keyed by their padded shapes only, so everything the gathers do not overwrite (padding rows and
# padding biases) is re-established on EVERY call: a smaller problem that lands in the same padded shape must not
# see the previous call's operands behind its own.
_BUFFERS: dict = {}


def _buffer(key, shape, dtype, device):
    full_key = (key, tuple(shape), dtype, device)
    buf = _BUFFERS.get(full_key)
    if buf is None:
        if len(_BUFFERS) > 64:
            _BUFFERS.clear()
        buf = _BUFFERS[full_key] = torch.empty(shape, dtype=dtype, device=device)
    return buf


def _pick_chunks(rows: int, world: int, chunks: Optional[int], own_first: bool = False) -> int:
    """Pieces the operand exchange is cut into (each piece: one all-gather + the scan launches behind it).  With the
    rank's own rows scanned first the whole exchange travels under that scan, so one piece is the default; without,
    the first piece is exposed and the exchange is cut in up to four."""
    if chunks is not None:
        if chunks < 1 or rows % chunks:
            raise ValueError(f"chunks={chunks} must divide the {rows} rows of a shard")
        return chunks
    if own_first:
        return 1
    for s in (4, 2):
        if rows % s == 0 and (rows // s) * world >= 16384:
            return s
    return 1


def _overlapped_simtopk(x_local, n_total, lo, hi, world, *, metric, lam, k, exclude_self, operand, group, return_stats,
                        chunks=None, col_splits=0, own_first=None, query_order="auto"):
    """Pipelined phase path.  Each rank prepares the 16-bit operands of its own rows and the ranks
    exchange them in S chunks (chunk c = rows [c*rows/S, (c+1)*rows/S) of every rank) by asynchronous all-gathers.

    own_first (default for world > 1; SURVEY.md §8(e) "scan the local shard first"): the first scan launch takes the
    rank's OWN rows as its columns — they are local, nothing is waited for — and every gathered chunk is scanned as the
    two column ranges either side of the rank's own segment (ranks below it, ranks above it), each behind the chunk's
    arrival event.  The whole exchange travels under the scan of the own rows.
    own_first=False: every column is scanned out of the gathered chunks (S launches, the first chunk exposed).

    The f32 all-gather, needed only by the exact re-rank, is issued last and is waited for between the last scan launch
    and the re-rank."""
    from . import ops
    dev = x_local.device
    rows, d = x_local.shape
    dp = ops.padded_dim(d)
    me = dist.get_rank(group)
    if own_first is None:
        own_first = world > 1
    S = _pick_chunks(rows, world, chunks, own_first)
    seg = rows // S
    rows_pad = (rows + 255) // 256 * 256
    z16 = torch.float16 if operand == "f16" else torch.bfloat16
    # ---- local phase -------------------------------------------------------------------------------------------
    pack_l = _buffer("pack", (5, rows_pad), torch.float32, dev)                    # scal, zn, rn, un, cb
    if rows_pad != rows:
        pack_l[:, rows:].zero_()
    maxn = torch.zeros((1,), dtype=torch.float32, device=dev)
    ops.row_scalars(x_local, metric, pack_l[0], maxn)
    if metric != "cosine":
        _allreduce_max(maxn, group)                      # the common power-of-two scale needs the global maximum
    z_buf = _buffer("z_l", (rows_pad + 256, dp), z16, dev)                        # + the slack a scan panel needs behind it
    z_buf[rows:].zero_()
    z_l = z_buf[:rows_pad]
    send = _buffer("send", (5 * rows + 4,), torch.float32, dev)                   # per-row scalars + this shard's maxima
    max4 = send[5 * rows:]
    max4.zero_()
    ops.prep_rows(x_local, metric, operand, pack_l[0], maxn, z_l, pack_l[1], pack_l[2], pack_l[3], pack_l[4], max4)
    send[:5 * rows].view(5, rows).copy_(pack_l[:, :rows])
    # ---- exchanges, in the order the scan needs them (collectives of a group complete in issue order) ------------
    recv = _buffer("recv", (world, 5 * rows + 4), torch.float32, dev)
    _gather_into(recv, send, group)
    m_c = world * seg
    m_pad = (m_c + 255) // 256 * 256
    zc = _buffer("zc", (S, m_pad + 512, dp), z16, dev)
    zc[:, m_c:].zero_()                                  # padding rows (never touched by the gathers): zero operands ...
    side_stream = _BUFFERS.get(("side_stream", dev))
    if side_stream is None:
        side_stream = _BUFFERS[("side_stream", dev)] = torch.cuda.Stream(device=dev)
    events = [_gather_event(zc[c, :m_c], z_l[c * seg:(c + 1) * seg], group, side_stream) for c in range(S)]
    full = _buffer("full", (n_total, d), x_local.dtype, dev)
    ev_full = _gather_event(full, x_local, group, side_stream)
    # ---- candidate-side scalars out of the small gather (overlaps the chunk exchange) ----------------------------
    max_all = recv[:, 5 * rows:].max(dim=0).values.contiguous()
    per_row = recv[:, :5 * rows].view(world, 5, rows)
    c_scal = per_row[:, 0].reshape(n_total)
    bias = per_row[:, 4].reshape(world, S, seg)          # [rank, chunk, row of the chunk]
    ninf = float("-inf")
    panels = []
    if own_first:
        cb_own = _buffer("cb_own", (rows_pad + 256,), torch.float32, dev)
        cb_own[rows:].fill_(ninf)                        # ... and -inf biases, the scan's only column mask
        cb_own[:rows].copy_(pack_l[4, :rows])
        panels.append(dict(Z=z_buf, cb=cb_own, m=rows, m_pad=rows_pad, seg_len=0, seg_stride=0, id_base=me * rows, event=None))
        below, above = me * seg, (world - 1 - me) * seg  # columns of a gathered chunk either side of the own segment
        cb_lo = _buffer("cb_lo", (S, m_pad + 256), torch.float32, dev)
        cb_hi = _buffer("cb_hi", (S, m_pad + 256), torch.float32, dev)
        if below:
            cb_lo[:, below:].fill_(ninf)
            cb_lo[:, :below] = bias[:me].permute(1, 0, 2).reshape(S, below)
        if above:
            cb_hi[:, above:].fill_(ninf)
            cb_hi[:, :above] = bias[me + 1:].permute(1, 0, 2).reshape(S, above)
        for c in range(S):
            if below:
                panels.append(dict(Z=zc[c], cb=cb_lo[c], m=below, m_pad=(below + 255) // 256 * 256, seg_len=seg, seg_stride=rows,
                                   id_base=c * seg, event=events[c]))
            if above:
                panels.append(dict(Z=zc[c, (me + 1) * seg:], cb=cb_hi[c], m=above, m_pad=(above + 255) // 256 * 256, seg_len=seg,
                                   seg_stride=rows, id_base=(me + 1) * rows + c * seg, event=events[c]))
    else:
        cb = _buffer("cb", (S, m_pad + 256), torch.float32, dev)
        cb[:, m_c:].fill_(ninf)
        cb[:, :m_c] = bias.permute(1, 0, 2).reshape(S, m_c)
        panels = [dict(Z=zc[c], cb=cb[c], m=m_c, m_pad=m_pad, seg_len=seg, seg_stride=rows, id_base=c * seg, event=events[c])
                  for c in range(S)]
    q = dict(Z=z_l, scal=pack_l[0], zn=pack_l[1], rn=pack_l[2], un=pack_l[3], cb=pack_l[4])
    out = ops.simtopk_panels(x_local, full, q, c_scal, panels, max_all, operand=operand, metric=metric, lam=lam, k=k,
                             exclude_self=exclude_self, row_offset=lo, col_offset=0, wait_event=ev_full,
                             col_splits=col_splits, profile=return_stats, return_stats=return_stats, query_order=query_order)
    if return_stats:
        out[2]["panels"] = len(panels)
        out[2]["own_first"] = bool(own_first)
    return out


def pick_driver(x_local: torch.Tensor, n_total: int, world: int, *, metric="cosine", k: int = 5, exclude_self: bool = True,
                precision: str = "auto", overlap: Optional[bool] = None, op: Optional[Callable] = None) -> str:
    """Which of the two drivers sharded_simtopk runs for these arguments: "pipelined" or "simple".  Pure shape /
    argument logic (no collective, no device work), so every rank reaches the same answer on its own."""
    if op is not None or overlap is False or not x_local.is_cuda or not isinstance(metric, str):
        return "simple"
    if world <= 1 and overlap is not True:     # overlap=True: the pipelined driver even for one rank (its collectives,
        return "simple"                        # events and panel hand-off run on a one-GPU box: tests/test_gpu_distributed.py)
    from . import ops as _ops
    if (n_total % world) == 0 and precision in ("auto", "fast", "fast_bf16") \
            and _ops.fast_scan_supported(x_local.shape[1], k, exclude_self) \
            and x_local.dtype in (torch.float32, torch.float16, torch.bfloat16):
        return "pipelined"
    return "simple"


def sharded_simtopk(x_local: torch.Tensor, n_total: int, *, metric="cosine", lam: float = 1.0, k: int = 5,
                    exclude_self: bool = True, precision: str = "auto", group=None, gather_output: bool = False,
                    op: Optional[Callable] = None, return_stats: bool = False, overlap: Optional[bool] = None,
                    chunks: Optional[int] = None, col_splits: int = 0, own_first: Optional[bool] = None,
                    query_order: str = "auto"):
    """Top-k of every local row against ALL n_total rows.

    query_order ("auto" / "off" / "on"): the order in which a rank's scan takes ITS rows (csrc/mmf_order.hip; near-duplicate rows
    next to each other, decided from a probe of the rank's own rows from 32768 rows per rank; never changes a result).

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
    driver = pick_driver(x_local, n_total, world, metric=metric, k=k, exclude_self=exclude_self, precision=precision,
                         overlap=overlap, op=op)
    if driver == "pipelined":
        out = _overlapped_simtopk(x_local.contiguous(), n_total, lo, hi, world, metric=metric, lam=lam, k=k,
                                  exclude_self=exclude_self, operand="bf16" if precision == "fast_bf16" else "f16",
                                  group=group, return_stats=return_stats, chunks=chunks, col_splits=col_splits,
                                  own_first=own_first, query_order=query_order)
        idx, val = out[0], out[1]
        if gather_output:
            idx = all_gather_rows(idx, n_total, group)
            val = all_gather_rows(val, n_total, group)
        if return_stats:
            out[2]["driver"] = "pipelined"
            return idx, val, out[2]
        return idx, val
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
        out = op(x_rows, full, precision=precision, return_stats=return_stats, profile=return_stats, query_order=query_order, **kw)
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
        if stats is not None:
            stats["driver"] = "simple"
        return idx, val, stats
    return idx, val
