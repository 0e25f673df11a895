#!/usr/bin/env python3
"""bench.py — similarity-pairs/sec of the fused similarity + per-row top-k (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over the whole synthetic workload: N = 262144 rows, d = 512,
cosine, k = 5, self excluded (BASELINE.json: "N x N cosine+top-k, N=262144 d=512; 1/2/4/8 GPU").
With P ranks the N rows are sharded P ways (strong scaling: total work is fixed); a step then includes the
exchange of the shards (multimodal-fusion_amd/distributed.py) and every rank scans all N columns for its rows.
value = N*N pairs / max-over-ranks step time.  Inputs are resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline       the scan kernel against the MFMA peak of the pipe it ran on (duration from HIP events recorded
                 inside the library on the launch stream)
  roofline_bf16  N = 1 only: the same workload once more with bf16 scan operands (MMF_PREC_FAST_BF16), untimed
                 in `value` — north_star words its target on the bf16 pipe
  cpu_baseline   the CPU oracle on a bounded sample of the same workload, rank 0, N = 1 only
  per_rank       N > 1: [min, max] over ranks of the step and of its parts (exposed_comm_ms = what the scan stream
                 spent waiting for operand chunks to arrive)
and in `config`: `driver` (which multi-GPU driver ran) and `self_check` (after the timed loop every rank re-computes
256 of its rows through the single-call path on the gathered matrix and compares bit for bit; a mismatch exits 1).
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
DTYPE_NAME = {1: "f32", 2: "f16 MFMA scan + f32 exact re-rank", 3: "bf16 MFMA scan + f32 exact re-rank"}
CLUSTERS, SIGMA = 2048, 0.03    # --data clustered: near-duplicate patches (what slide embeddings look like)


def make_rows(lo: int, hi: int, d: int, device, block: int = 4096, data: str = "gaussian"):
    """Rows lo..hi of the synthetic matrix, generated per 4096-row block from a seed that depends on the block index
    only, so every shard count sees identical data.
    gaussian : randn, L2-normalised
    clustered: a cluster centre (one of CLUSTERS unit vectors) + SIGMA-sized noise, L2-normalised — rows of a
               cluster sit within the 16-bit scan's error margin of each other (DESIGN.md §4.1 "near-duplicate data")"""
    import torch
    out = torch.empty((hi - lo, d), dtype=torch.float32, device=device)
    centers = None
    if data == "clustered":
        g = torch.Generator(device=device).manual_seed(77)
        centers = torch.randn((CLUSTERS, d), generator=g, device=device, dtype=torch.float32)
        centers = centers / centers.norm(dim=1, keepdim=True)
    b0 = lo // block
    b1 = (hi + block - 1) // block
    for b in range(b0, b1):
        g = torch.Generator(device=device).manual_seed(1234 + b)
        blk = torch.randn((block, d), generator=g, device=device, dtype=torch.float32)
        if centers is not None:
            assign = torch.randint(0, CLUSTERS, (block,), generator=g, device=device)
            blk = centers[assign] + (SIGMA / d ** 0.5) * blk
        blk = blk / blk.norm(dim=1, keepdim=True)
        s, e = max(lo, b * block), min(hi, (b + 1) * block)
        out[s - lo:e - lo] = blk[s - b * block:e - b * block]
    return out


def cpu_baseline(n: int, d: int, k: int, metric: str, device, data: str):
    """The CPU oracle (canonical C restatement, OpenMP) on a bounded sample: R query rows of the
    workload against its first C columns, sized for ~15 s of CPU work."""
    import oracle
    cols = min(n, 65536)
    Y = make_rows(0, cols, d, device, data=data).cpu().numpy()
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
    ap.add_argument("--data", default="gaussian", choices=["gaussian", "clustered"])
    ap.add_argument("--query-order", default="auto", choices=["auto", "off", "on"],
                    help="order in which the scan takes the query rows (csrc/mmf_order.hip); auto = the library's own decision")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-leg", action="store_true")
    ap.add_argument("--simple-driver", action="store_true", help="N > 1: one all-gather of the f32 shard, then one scan")
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
    x_local = make_rows(lo, hi, d, device, data=args.data)
    torch.cuda.synchronize()

    # Which driver runs is a pure function of the arguments (no probe, no fallback): if it raises, the job fails.
    overlap = False if args.simple_driver else None
    driver = dmod.pick_driver(x_local, n, world, metric=args.metric, k=k, exclude_self=True, precision=args.precision,
                              overlap=overlap) if world > 1 else "single"

    def step(profile: bool, precision: str = args.precision):
        return dmod.sharded_simtopk(x_local, n, metric=args.metric, k=k, exclude_self=True, precision=precision,
                                    return_stats=profile, overlap=overlap, query_order=args.query_order)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(precision: str):
        for _ in range(args.warmup):
            step(False, precision)
        per_step, st, out = [], None, None
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            i, v, st = step(True, precision)
            per_step.append(st)
            out = (i, v)
        fence()
        return time.perf_counter() - t0, per_step, out

    elapsed, per_step, (out_i, out_v) = timed(args.precision)
    stats = per_step[-1]
    mean = lambda key: sum(float(s.get(key, 0.0)) for s in per_step) / len(per_step)   # noqa: E731
    local_ms = [1e3 * elapsed / args.steps, mean("prep_ms"), mean("scan_ms"), mean("scan_wait_ms"), mean("rerank_ms"),
                mean("fallback_ms")]
    per_rank, check = None, None
    if world > 1:
        red_dev = device if args.backend == "nccl" else torch.device("cpu")   # gloo reduces host tensors

        def reduce(values, op):
            t = torch.tensor(values, dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=op)
            return [float(x) for x in t.tolist()]
        elapsed = reduce([elapsed], dist.ReduceOp.MAX)[0]
        # ---- self-check (untimed): 256 of this rank's rows through the single-call path on the gathered matrix ----
        full = dmod.all_gather_rows(x_local, n, None)
        nchk = min(256, hi - lo)
        ci, cv = mmf.simtopk(full[lo:lo + nchk], full, metric=args.metric, k=k, exclude_self=True, row_offset=lo,
                             precision=args.precision)
        good = bool(torch.equal(ci, out_i[:nchk])) and bool(torch.equal(cv, out_v[:nchk]))
        del full
        all_good = reduce([1.0 if good else 0.0], dist.ReduceOp.MIN)[0] == 1.0
        mn, mx = reduce(local_ms, dist.ReduceOp.MIN), reduce(local_ms, dist.ReduceOp.MAX)
        names = ("step_ms", "prep_ms", "scan_ms", "exposed_comm_ms", "rerank_ms", "fallback_ms")
        per_rank = {nm: [mn[j], mx[j]] for j, nm in enumerate(names)}
        check = {"rows_per_rank": nchk, "ok": all_good,
                 "against": "mmf_simtopk(full[lo:lo+256], full, row_offset=lo), indices and scores bit for bit"}
        if not good:
            print(f"[bench] rank {rank}: sharded result differs from the single-call path on its first {nchk} rows",
                  file=sys.stderr, flush=True)

    def roofline_of(per_step_stats, rows_local):
        st = per_step_stats[-1]
        prec = st["precision_used"]
        scan_ms = sum(s["scan_ms"] - s.get("scan_wait_ms", 0.0) for s in per_step_stats) / len(per_step_stats)
        flops = 2.0 * rows_local * n * d                                # 2*d flop per pair
        achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
        peak = PEAK_TFLOPS.get(prec, 157.3)
        return prec, {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                      "kernel": SCAN_NAME[prec], "kernel_ms": scan_ms}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        pairs = float(n) * float(n)
        prec, roof = roofline_of(per_step, hi - lo)
        # HBM-side bytes per scan launch come from separate rocprofv3 --pmc passes (profiles/README.md), not from this run
        roof["traffic"], roof["traffic_source"] = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(f"N={n},d={d},gpus={world},prec={prec},data={args.data}") or \
                (tj.get(f"N={n},d={d},gpus={world},prec={prec}") if args.data == "gaussian" else None)
            if ent:
                roof["traffic"] = ent.get("hbm_bytes_per_launch")
                roof["traffic_source"] = "offline rocprofv3 --pmc passes of this command, " + str(ent.get("round", "profiles/"))
        except (OSError, ValueError):
            pass
        roof.update({"prep_ms": stats["prep_ms"], "order_ms": stats.get("order_ms", 0.0), "rerank_ms": stats["rerank_ms"],
                     "fallback_ms": stats["fallback_ms"]})
        par = {"single": "single GPU",
               "simple": f"row-shard x{world}; one all-gather of the f32 shard",
               "pipelined": f"row-shard x{world}; own rows scanned first, 16-bit operands of the other ranks all-gathered under that scan, f32 shard under all of it"}[driver]
        line = {
            "metric": "similarity-pairs/sec (NxN cosine+top-k)", "value": pairs / (elapsed / args.steps),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE_NAME[prec], "data": "synthetic",
            "config": {"workload": f"N={n} d={d} single-modality {args.metric} + top-{k}, self excluded, f32 features, "
                                   f"{'unit-norm Gaussian rows' if args.data == 'gaussian' else f'clustered rows ({CLUSTERS} clusters, sigma {SIGMA})'}",
                       "rows_per_rank": hi - lo, "parallelism": par, "driver": driver,
                       "scan_kernel": SCAN_NAME[prec], "col_splits": stats["col_splits"],
                       "scan_grid": stats["scan_grid"], "fallback_rows": stats["fallback_rows"],
                       "overflow_rows": stats["overflow_rows"],
                       "candidates_per_row": stats["candidates"] / max(1, hi - lo),
                       # the scan's query order (csrc/mmf_order.hip): decided per call from a probe of the rows
                       "query_order": bool(stats.get("query_order", 0)), "near_duplicate_rows_estimate": stats.get("near_rows", -1)},
            "roofline": roof,
        }
        if check is not None:
            line["config"]["self_check"] = check
            line["per_rank"] = per_rank
        print_line = line
    if world == 1 and not args.no_bf16_leg and args.precision in ("auto", "fast"):
        _, per_step_b, _ = timed("fast_bf16")
        stb = per_step_b[-1]
        if stb["precision_used"] == 3:
            _, rb = roofline_of(per_step_b, hi - lo)
            rb.update({"candidates_per_row": stb["candidates"] / max(1, hi - lo), "fallback_rows": stb["fallback_rows"],
                       "note": "same workload with bf16 scan operands, separate untimed-in-value leg"})
            print_line["roofline_bf16"] = rb
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            print_line["cpu_baseline"] = cpu_baseline(n, d, k, args.metric, device, args.data)
        print(json.dumps(print_line), flush=True)
    bad = check is not None and not check["ok"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
