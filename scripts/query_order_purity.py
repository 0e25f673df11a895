"""How pure are the scan's 32-row waves under the pivot order?  (torch emulation of csrc/mmf_order.hip's key on the clustered
bench workload, ground-truth clusters from the generator's centres)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import multimodal_fusion_amd as mmf
dev = torch.device("cuda", 0)
N, d = 262144, 512
X = bench.make_rows(0, N, d, dev, data="clustered")
g = torch.Generator(device=dev).manual_seed(77)
centers = torch.randn((bench.CLUSTERS, d), generator=g, device=dev)
centers = centers / centers.norm(dim=1, keepdim=True)
cl = torch.cat([(X[i:i + 16384] @ centers.T).argmax(1) for i in range(0, N, 16384)])
Xh = X.half().float()


def waves(order):
    c = cl[order].view(-1, 32)
    s, _ = torch.sort(c, dim=1)
    return float(((s[:, 1:] != s[:, :-1]).sum(1) + 1).float().mean())


print(f"generator order: {waves(torch.arange(N, device=dev)):.2f} clusters per wave; cluster order: {waves(torch.argsort(cl, stable=True)):.2f}")
for P in (64, 128, 256, 512, 1024, 2048):
    piv = Xh[(torch.arange(P, device=dev) * N) // P]
    S = Xh @ piv.T
    best, bp = S.max(1)
    key = bp.double() * 4 + (best.double() + 1)
    o1 = torch.argsort(key)
    top2 = S.topk(2, dim=1)
    key2 = (top2.indices[:, 0].double() * P + top2.indices[:, 1].double()) * 4 + (top2.values[:, 0].double() + 1)
    o2 = torch.argsort(key2)
    print(f"P = {P:5d}: (pivot, cosine) {waves(o1):.2f}   (pivot, second pivot, cosine) {waves(o2):.2f}", flush=True)

i, v, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", query_order="on", return_stats=True, profile=True)
perm = mmf.ops.last_query_order(N).to(dev).long()
assert torch.equal(torch.sort(perm).values, torch.arange(N, device=dev))
print(f"the library's order (csrc/mmf_order.hip): {waves(perm):.2f} clusters per wave; scan {st['scan_ms']:.2f} ms")
