"""Kernel-by-kernel view of the small cluster-shaped steps (run under rocprofv3 --kernel-trace)."""
import sys, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
ops = mmf.ops
dev = torch.device('cuda')
N2, k, d = 262144, 5, 512
g = torch.Generator(device=dev).manual_seed(3)
X = torch.randn((N2, d), generator=g, device=dev)
X = X / X.norm(dim=1, keepdim=True)
idx, _ = mmf.simtopk(X, metric="cosine", k=k)
lab = torch.randint(0, 100, (N2,), device=dev)
for _ in range(3):
    ops.knn_pairs(idx)
    seg = ops.segment_sort(lab, 100)
    ops.segment_mean(X, seg)
torch.cuda.synchronize()
