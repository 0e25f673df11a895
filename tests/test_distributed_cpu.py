"""world_size-2 gloo test of the row-sharded driver (multimodal-fusion_amd/distributed.py).

The collective plumbing (one all-gather of the feature shard, global row offsets, optional gather of
the outputs) is exercised for real over gloo; the per-rank device op is replaced by the CPU oracle
(allowed here: this is a test).  Sharded == unsharded, bit for bit, including uneven shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_op(x_local, full, *, metric, lam, k, exclude_self, row_offset, col_offset):
    import oracle
    idx, val = oracle.simtopk(x_local.numpy(), full.numpy(), metric=metric, lam=lam, k=k, exclude_self=exclude_self,
                              row_offset=row_offset, col_offset=col_offset, nthreads=2)
    return torch.from_numpy(idx), torch.from_numpy(val)


def _worker(rank, world, port, n, d, k, metric, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    import multimodal_fusion_amd  # noqa: F401
    dmod = import_module("multimodal_fusion_amd.distributed")
    g = torch.Generator().manual_seed(4242)
    X = torch.randn(n, d, generator=g)
    lo, hi = dmod.shard_bounds(n, world, rank)
    idx, val = dmod.sharded_simtopk(X[lo:hi].contiguous(), n, metric=metric, k=k, exclude_self=True,
                                    gather_output=True, op=_oracle_op)
    own_i, own_v = dmod.sharded_simtopk(X[lo:hi].contiguous(), n, metric=metric, k=k, exclude_self=True, op=_oracle_op)
    assert torch.equal(own_i, idx[lo:hi]) and torch.equal(own_v, val[lo:hi])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx.numpy(), val=val.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(96, 2), (101, 2), (67, 3)])
def test_sharded_equals_unsharded_gloo(tmp_path, n, world):
    import oracle
    d, k, metric = 24, 4, "cosine"
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, d, k, metric, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(4242)
    X = torch.randn(n, d, generator=g).numpy()
    ridx, rval = oracle.simtopk(X, metric=metric, k=k)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["idx"], ridx) and np.array_equal(z["val"], rval)


def test_shard_bounds_partition():
    from importlib import import_module
    sys.path.insert(0, ROOT)
    import multimodal_fusion_amd  # noqa: F401
    dmod = import_module("multimodal_fusion_amd.distributed")
    for n in (0, 1, 7, 64, 262144, 262145):
        for w in (1, 2, 3, 8):
            b = [dmod.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_chunk_layout_of_the_pipelined_exchange_maps_back_to_global_columns():
    """The pipelined driver all-gathers chunk c = rows [c*rows/S, (c+1)*rows/S) of every rank; rank-major
    concatenation makes column i of that panel global column id_base + (i // seg_len) * seg_stride + i % seg_len
    (include/mmf_hg.h, mmf_panel).  Check the formula against an explicit gather, and that the panels tile [0, N)."""
    import importlib
    dmod = importlib.import_module("multimodal_fusion_amd.distributed")
    for world, rows, S in ((8, 64, 4), (4, 30, 2), (3, 7, 1), (2, 12, 3)):
        n = world * rows
        seg = rows // S
        owner_rows = [np.arange(*dmod.shard_bounds(n, world, r)) for r in range(world)]
        seen = []
        for c in range(S):
            gathered = np.concatenate([owner_rows[r][c * seg:(c + 1) * seg] for r in range(world)])   # all_gather_into_tensor
            i = np.arange(world * seg)
            mapped = c * seg + (i // seg) * rows + i % seg
            assert np.array_equal(mapped, gathered), (world, rows, S, c)
            seen.append(mapped)
        assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(n))
    # own rows first (the default for world > 1): the rank's own rows, then each gathered chunk as the column ranges either
    # side of the own segment — panels of rank `me` tile [0, N) without touching its own rows twice
    for world, rows, S in ((8, 64, 1), (8, 64, 4), (4, 30, 2), (2, 12, 3), (3, 7, 1)):
        n = world * rows
        seg = rows // S
        for me in range(world):
            seen = [me * rows + np.arange(rows)]
            for c in range(S):
                gathered = np.concatenate([r * rows + c * seg + np.arange(seg) for r in range(world)])
                i = np.arange(me * seg)
                lo_ids = c * seg + (i // seg) * rows + i % seg
                assert np.array_equal(lo_ids, gathered[:me * seg])
                i = np.arange((world - 1 - me) * seg)
                hi_ids = (me + 1) * rows + c * seg + (i // seg) * rows + i % seg
                assert np.array_equal(hi_ids, gathered[(me + 1) * seg:])
                seen += [lo_ids, hi_ids]
            assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(n)), (world, rows, S, me)
    assert dmod._pick_chunks(32768, 8, None) == 4 and dmod._pick_chunks(1000, 2, None) == 1
    assert dmod._pick_chunks(32768, 8, None, own_first=True) == 1 and dmod._pick_chunks(32768, 8, 4, own_first=True) == 4
    assert dmod._pick_chunks(32768, 8, 2) == 2
    with pytest.raises(ValueError):
        dmod._pick_chunks(1000, 2, 3)
