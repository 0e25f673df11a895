#!/usr/bin/env python3
"""bench.py — similarity-pairs/sec of the fused similarity + per-row top-k (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over the whole synthetic workload: N = 262144 rows, d = 512,
cosine, k = 5, self excluded (BASELINE.json: "N x N cosine+top-k, N=262144 d=512; 1/2/4/8 GPU").
With P ranks the N rows are sharded P ways (strong scaling: total work is fixed), each step does ONE
RCCL all-gather of the f32 feature shard and then scans all N columns for the local rows.
value = N*N pairs / max-over-ranks step time.  Inputs are resident in HBM before the timed region.

Extra objects on the JSON line: `roofline` (the scan kernel against the MFMA peak of the pipe it ran
on, duration from HIP events recorded inside the library on the launch stream) and `cpu_baseline`
(the CPU oracle on a bounded sample of the same workload, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {1: 157.3, 2: 2500.0, 3: 2500.0}   # f32 MFMA; f16 / bf16 MFMA dense (MI355X_MICROARCH.md)
SCAN_NAME = {1: "scan_f32 (v_mfma_f32_32x32x2_f32)", 2: "scan_b16x<f16> (v_mfma_f32_16x16x32_f16)",
             3: "scan_b16x<bf16> (v_mfma_f32_16x16x32_bf16)"}


def make_rows(lo: int, hi: int, d: int, device, block: int = 4096):
    """Rows lo..hi of the synthetic matrix: randn per 4096-row block seeded by the block index, then
    L2-normalised, so every shard count sees identical data."""
    import torch
    out = torch.empty((hi - lo, d), dtype=torch.float32, device=device)
    b0 = lo // block
    b1 = (hi + block - 1) // block
    for b in range(b0, b1):
        g = torch.Generator(device=device).manual_seed(1234 + b)
        blk = torch.randn((block, d), generator=g, device=device, dtype=torch.float32)
        blk = blk / blk.norm(dim=1, keepdim=True)
        s, e = max(lo, b * block), min(hi, (b + 1) * block)
        out[s - lo:e - lo] = blk[s - b * block:e - b * block]
    return out


def cpu_baseline(n: int, d: int, k: int, metric: str, device):
    """The CPU oracle (canonical C restatement, OpenMP) on a bounded sample: R query rows of the
    workload against its first C columns, sized for ~15 s of CPU work."""
    import torch
    import oracle
    cols = min(n, 65536)
    Y = make_rows(0, cols, d, device).cpu().numpy()
    threads = oracle.num_threads()
    oracle.simtopk(Y[:64], Y[:4096], metric=metric, k=k, exclude_self=True)        # thread pool start-up
    rows, dt = 512, 0.0
    while True:                                                                     # grow until ~10 s
        t0 = time.perf_counter()
        oracle.simtopk(Y[:rows], Y, metric=metric, k=k, exclude_self=True)
        dt = time.perf_counter() - t0
        if dt >= 8.0 or rows >= cols:
            break
        rows = int(min(cols, max(rows * 2, rows * 10.0 / max(dt, 1e-3))))
        rows -= rows % 4
    return {"value": rows * cols / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{rows} query rows x {cols} columns of the same workload (d={d}, {metric}, k={k}), "
                      f"oracle/mmf_oracle.c with {threads} OpenMP threads, {dt:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=262144)
    ap.add_argument("--dim", dest="d", type=int, default=512)
    ap.add_argument("--topk", dest="k", type=int, default=5)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--precision", default="auto", choices=["auto", "exact", "fast", "fast_bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import multimodal_fusion_amd as mmf
    from importlib import import_module
    dmod = import_module("multimodal_fusion_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    n, d, k = args.n, args.d, args.k
    lo, hi = dmod.shard_bounds(n, world, rank)
    x_local = make_rows(lo, hi, d, device)
    torch.cuda.synchronize()

    driver = {"overlap": None}      # None: the pipelined driver where it applies (multimodal-fusion_amd/distributed.py)

    def step(profile: bool):
        return dmod.sharded_simtopk(x_local, n, metric=args.metric, k=k, exclude_self=True,
                                    precision=args.precision, return_stats=profile, overlap=driver["overlap"])

    if world > 1:
        # One untimed probe of the pipelined driver before anything is measured.  Every rank runs the same code on
        # the same shapes, so a failure raises on all of them; the job then continues on the simple driver (one
        # all-gather of the f32 shard) instead of dying without a number.  The all-reduce makes the choice common.
        ok = torch.ones(1, device=device)
        try:
            step(False)
            torch.cuda.synchronize()
        except Exception as exc:                                            # noqa: BLE001
            ok.zero_()
            if rank == 0:
                print(f"[bench] pipelined driver failed ({type(exc).__name__}: {exc}); using the simple driver", file=sys.stderr, flush=True)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            driver["overlap"] = False
    for _ in range(args.warmup):
        step(False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    scan_ms, stats = [], None
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, _, stats = step(True)
        scan_ms.append(stats["scan_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        pairs = float(n) * float(n)
        prec = stats["precision_used"]
        scan_avg_ms = sum(scan_ms) / len(scan_ms)
        flops_per_launch = 2.0 * (hi - lo) * n * d                    # 2*d flop per pair
        achieved = flops_per_launch / (scan_avg_ms * 1e-3) / 1e12 if scan_avg_ms > 0 else 0.0
        peak = PEAK_TFLOPS.get(prec, 157.3)
        traffic = None     # HBM-side bytes per scan launch, measured offline with rocprofv3 --pmc (profiles/README.md)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = tj.get(f"N={n},d={d},gpus={world},prec={prec}", {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        line = {
            "metric": "similarity-pairs/sec (NxN cosine+top-k)", "value": pairs / (elapsed / args.steps),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {1: "f32", 2: "f16 MFMA scan + f32 exact re-rank", 3: "bf16 MFMA scan + f32 exact re-rank"}[prec],
            "data": "synthetic",
            "config": {"workload": f"N={n} d={d} single-modality {args.metric} + top-{k}, self excluded, f32 features",
                       "rows_per_rank": hi - lo, "parallelism": ("single GPU" if world == 1 else
                                       f"row-shard x{world}; one all-gather of the f32 shard" if driver["overlap"] is False else
                                       f"row-shard x{world}; 16-bit operands all-gathered in chunks under the scan, f32 shard under all of it"),
                       "scan_kernel": SCAN_NAME[prec], "col_splits": stats["col_splits"],
                       "scan_grid": stats["scan_grid"], "fallback_rows": stats["fallback_rows"],
                       "candidates_per_row": stats["candidates"] / max(1, hi - lo)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "kernel": SCAN_NAME[prec], "kernel_ms": scan_avg_ms,
                         "prep_ms": stats["prep_ms"], "rerank_ms": stats["rerank_ms"], "fallback_ms": stats["fallback_ms"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, d, k, args.metric, device)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
