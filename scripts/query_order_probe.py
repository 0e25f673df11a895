"""Does the ORDER of the query rows matter to the 16-bit scan on near-duplicate data?  A wave scans for 32 consecutive queries and
enters its list code whenever ANY of them has a hit in the tile; if the 32 are near-duplicates of each other their hits coincide.
Probe: the clustered bench workload as a cross-similarity call (queries = a permutation of the rows, candidates = the rows), with
the queries in generator order, in cluster order (ground truth), and in random order."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
import bench
dev = torch.device("cuda", 0)
N, d = 262144, 512
X = bench.make_rows(0, N, d, dev, data="clustered")
# ground-truth cluster of every row: nearest centre
g = torch.Generator(device=dev).manual_seed(77)
centers = torch.randn((bench.CLUSTERS, d), generator=g, device=dev)
centers = centers / centers.norm(dim=1, keepdim=True)
cl = torch.cat([(X[i:i + 16384] @ centers.T).argmax(1) for i in range(0, N, 16384)])
orders = {"generator order": torch.arange(N, device=dev), "cluster order": torch.argsort(cl, stable=True),
          "random order": torch.randperm(N, device=dev)}
ref = None
for name, perm in orders.items():
    Q = X[perm].contiguous()
    for rep in range(3):
        i, v, st = mmf.simtopk(Q, X, metric="cosine", k=6, exclude_self=False, precision="fast", return_stats=True, profile=True)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(N, device=dev)
    res = i[inv]
    if ref is None:
        ref = res
    print(f"{name:16s} scan {st['scan_ms']:7.2f} ms  re-rank {st['rerank_ms']:6.2f} ms  candidates/row {st['candidates'] / N:6.1f}  same result {bool(torch.equal(res, ref))}", flush=True)
