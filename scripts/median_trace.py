"""Kernel-by-kernel view of the one-sweep median (run under rocprofv3 --kernel-trace --stats)."""
import sys, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
ops = mmf.ops
N = 16384
g = torch.Generator(device='cuda').manual_seed(1)
F = torch.randn((N, 512), generator=g, device='cuda') * 0.05
P = torch.rand((N, 2), generator=g, device='cuda') * 100
K = ops.sim_dense_combined(F, P, 0.5, 0.001)
for _ in range(5):
    ops.offdiag_lower_median(K)
torch.cuda.synchronize()
