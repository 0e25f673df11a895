"""Multi-rank rehearsal on ONE GPU (gloo + host staging): every rank runs the overlapped phase path and
the simple path and checks both against the unsharded result.  Launch with torch.distributed.run."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
from importlib import import_module
import multimodal_fusion_amd as mmf
dmod = import_module("multimodal_fusion_amd.distributed")
from bench import make_rows
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
for N, d, metric, dt in ((16384, 512, "cosine", torch.float32), (8192, 128, "neg_sq_l2", torch.float32),
                         (8192, 256, "cosine", torch.float16), (8192, 1024, "dot", torch.bfloat16)):
    X = (make_rows(0, N, d, dev) * (1.0 if metric == "cosine" else 2.5)).to(dt)
    ref_i, ref_v = mmf.simtopk(X, metric=metric, k=5)
    lo, hi = dmod.shard_bounds(N, world, rank)
    for overlap in (True, False):
        i, v = dmod.sharded_simtopk(X[lo:hi].clone(), N, metric=metric, k=5, overlap=overlap)
        ok = torch.equal(i, ref_i[lo:hi]) and torch.equal(v, ref_v[lo:hi])
        print(f"rank {rank}/{world} N={N} {metric} overlap={overlap}: {'OK' if ok else 'MISMATCH'}", flush=True)
        assert ok
    gi, gv = dmod.sharded_simtopk(X[lo:hi].clone(), N, metric=metric, k=5, gather_output=True)
    assert torch.equal(gi, ref_i) and torch.equal(gv, ref_v)
dist.barrier()
dist.destroy_process_group()
