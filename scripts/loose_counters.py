"""Event counters (MMF_SCAN_DEBUG=72) and timing-only ablations of the 16-bit scan on clusters of a given noise norm (see
scripts/query_order_loose.py).  scripts/loose_counters.py [noise norm, default 0.1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
dev = torch.device("cuda", 0)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
N, d, C = 262144, 512, 2048
g = torch.Generator(device=dev).manual_seed(11)
centers = torch.randn((C, d), generator=g, device=dev)
centers = centers / centers.norm(dim=1, keepdim=True)
assign = torch.randint(0, C, (N,), generator=g, device=dev)
X = centers[assign] + (noise / d ** 0.5) * torch.randn((N, d), generator=g, device=dev)
X = X / X.norm(dim=1, keepdim=True)
for name, bits in (("production", None), ("instrumented", 64), ("counters", 72), ("no overflow-list stores", 192), ("no list code", 65)):
    if bits is None:
        os.environ.pop("MMF_SCAN_DEBUG", None)
    else:
        os.environ["MMF_SCAN_DEBUG"] = str(bits)
    for rep in range(2):
        _, _, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", profile=True, return_stats=True)
    print(f"noise {noise}: {name:24s} scan {st['scan_ms']:8.2f} ms  ordered {st['query_order']}  candidates/row {st['candidates'] / N:.1f}  flagged {st['fallback_rows']}", flush=True)
