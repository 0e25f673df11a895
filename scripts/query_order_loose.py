"""Query order on LOOSER clusters than the bench's (noise norm 0.03 -> cosines inside a cluster 0.999): noise norms 0.1 .. 1.0, i.e.
cosines inside a cluster 0.99 .. 0.5 — the margin band is no longer the whole cluster, but the rows of a cluster still share their
neighbours.  scan time with the order off / on, and what AUTO decides."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
dev = torch.device("cuda", 0)
N, d, C = 262144, 512, 2048
g = torch.Generator(device=dev).manual_seed(11)
centers = torch.randn((C, d), generator=g, device=dev)
centers = centers / centers.norm(dim=1, keepdim=True)
assign = torch.randint(0, C, (N,), generator=g, device=dev)
for noise in ([float(x) for x in sys.argv[1:]] or [0.1, 0.3, 0.5, 1.0]):
    X = centers[assign] + (noise / d ** 0.5) * torch.randn((N, d), generator=g, device=dev)
    X = X / X.norm(dim=1, keepdim=True)
    out = {}
    for mode in ("off", "on", "auto"):
        for rep in range(2):
            i, v, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", return_stats=True, profile=True, query_order=mode)
        out[mode] = (i, v, st)
    same = all(torch.equal(out[m][0], out["off"][0]) and torch.equal(out[m][1], out["off"][1]) for m in ("on", "auto"))
    s = {m: out[m][2] for m in out}
    print(f"noise norm {noise:4.1f} (cosine inside a cluster {1 / (1 + noise * noise):.3f}): scan off {s['off']['scan_ms']:6.2f} ms  on {s['on']['scan_ms']:6.2f} ms  "
          f"re-rank off {s['off']['rerank_ms']:5.2f} on {s['on']['rerank_ms']:5.2f}  candidates/row {s['on']['candidates'] / N:6.1f}  "
          f"auto: near rows {s['auto']['near_rows']} ordered {s['auto']['query_order']}  same bits {same}", flush=True)
