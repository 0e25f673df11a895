"""The RCCL code path of the pipelined driver on a ONE-GPU box: backend "nccl" (= RCCL on ROCm) with world_size 1.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P scripts/rccl_one_rank.py

What runs on real RCCL here and never runs under the gloo rehearsal (scripts/rehearse_shards.py):
`dist.all_gather_into_tensor(..., async_op=True)` on the backend's stream, `work.wait()` on the side stream, the event
recorded behind it and handed to the library as mmf_panel.ready_event / mmf_simtopk_opts.select_wait_event, the
blocking small gather and the all-reduce(MAX) of the L2 metrics.  With one rank the gathered chunk IS the rank's own
rows, so the layout without "own rows first" is forced: every column is scanned out of RCCL's output buffers.
Results must equal mmf.simtopk bit for bit.  tests/test_gpu_distributed.py runs this under pytest -m gpu."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module   # noqa: E402

import multimodal_fusion_amd as mmf   # noqa: E402
from bench import make_rows            # noqa: E402

dmod = import_module("multimodal_fusion_amd.distributed")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
for N, d, metric, dt, S in ((32768, 512, "cosine", torch.float32, 4), (16384, 256, "neg_sq_l2", torch.float32, 2),
                            (16384, 128, "rbf", torch.float16, 1)):
    X = (make_rows(0, N, d, dev) * (1.0 if metric == "cosine" else 2.0)).to(dt)
    ref_i, ref_v = mmf.simtopk(X, metric=metric, lam=0.5, k=5)
    for rep in range(2):                        # twice: the cached exchange buffers are reused
        i, v, st = dmod.sharded_simtopk(X.clone(), N, metric=metric, lam=0.5, k=5, overlap=True, own_first=False, chunks=S,
                                        return_stats=True)
        assert st["driver"] == "pipelined" and st["panels"] == S and st["own_first"] is False, st
        ok = torch.equal(i, ref_i) and (torch.allclose(v, ref_v, rtol=0, atol=1e-5) if metric == "rbf" else torch.equal(v, ref_v))
        print(f"nccl world=1 N={N} d={d} {metric} chunks={S} rep={rep}: panels={st['panels']} wait={st.get('scan_wait_ms', 0.0):.3f} ms "
              f"{'OK' if ok else 'MISMATCH'}", flush=True)
        assert ok
    # own rows first at world 1: one panel, the gathers still run (and the re-rank waits for the f32 one)
    i, v, st = dmod.sharded_simtopk(X.clone(), N, metric=metric, lam=0.5, k=5, overlap=True, own_first=True, return_stats=True)
    assert st["panels"] == 1 and torch.equal(i, ref_i)
    # the simple driver's collective on RCCL
    gi, gv = dmod.sharded_simtopk(X.clone(), N, metric=metric, lam=0.5, k=5, overlap=False, gather_output=True)
    assert torch.equal(gi, ref_i)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL ONE RANK OK", flush=True)
