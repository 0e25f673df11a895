"""A/B of library variants in ONE process, interleaved rounds (boxes of the pool differ by a few per cent, and so do
separate processes): scripts/ab_inproc.py [--rows N] [--rounds R] [--prec fast|fast_bf16] base nw4 early ...
A name maps to multimodal-fusion_amd/libmmf_hg_<name>.so ("base" = the shipped library).  Every variant's result is
compared with the first variant's, bit for bit."""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_fusion_amd as mmf   # noqa: E402
from bench import make_rows            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=262144)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--prec", default="fast")
ap.add_argument("--data", default="gaussian")
ap.add_argument("--topk", type=int, default=5)
ap.add_argument("--half", action="store_true", help="fp16 features")
ap.add_argument("--noise", type=float, default=0.0, help="> 0: 2048 clusters with this noise norm (scripts/query_order_loose.py) instead of --data")
ap.add_argument("names", nargs="+")
a = ap.parse_args()
dev = torch.device("cuda", 0)
X = make_rows(0, a.rows, a.dim, dev, data=a.data)
if a.noise > 0:
    g = torch.Generator(device=dev).manual_seed(11)
    centers = torch.randn((2048, a.dim), generator=g, device=dev)
    centers = centers / centers.norm(dim=1, keepdim=True)
    X = centers[torch.randint(0, 2048, (a.rows,), generator=g, device=dev)] + (a.noise / a.dim ** 0.5) * torch.randn((a.rows, a.dim), generator=g, device=dev)
    X = X / X.norm(dim=1, keepdim=True)
if a.half:
    X = X.half()


def use(name):
    path = os.path.join(ROOT, "multimodal-fusion_amd", "libmmf_hg.so" if name == "base" else f"libmmf_hg_{name}.so")
    lib = libs.get(name)
    mmf._lib._lib = lib
    mmf._lib.SO_PATH = path
    if lib is None:
        libs[name] = mmf._lib.lib()


libs, times, rerank, ref = {}, {n: [] for n in a.names}, {n: [] for n in a.names}, None
cands, flagged = {}, {}
for r in range(a.rounds + 1):
    for n in a.names:
        use(n)
        i, v, st = mmf.simtopk(X, metric="cosine", k=a.topk, precision=a.prec, profile=True, return_stats=True)
        if r == 0:                      # warm-up round: also the parity check
            if ref is None:
                ref = (i, v)
            else:
                assert torch.equal(i, ref[0]) and torch.equal(v, ref[1]), f"{n}: result differs from {a.names[0]}"
            continue
        cands[n], flagged[n] = st["candidates"] / a.rows, st["fallback_rows"]
        times[n].append(st["scan_ms"])
        rerank[n].append(st["rerank_ms"])
for n in a.names:
    t = times[n]
    print(f"{n:12s} scan_ms median {statistics.median(t):7.3f}  min {min(t):7.3f}  max {max(t):7.3f}   "
          f"frac(median) {2.0 * a.rows * a.rows * a.dim / (statistics.median(t) * 1e-3) / 2.5e15:.4f}   "
          f"rerank_ms median {statistics.median(rerank[n]):6.3f}   candidates/row {cands[n]:6.1f}  flagged {flagged[n]}", flush=True)
