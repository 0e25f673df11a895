"""build_weighted_hypergraph at an N whose K one would rather not keep: median + edges from recomputed row panels."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
ops = mmf.ops
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 512
g = torch.Generator(device='cuda').manual_seed(1)
F = torch.randn((N, d), generator=g, device='cuda') * 0.05
P = torch.rand((N, 2), generator=g, device='cuda')
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    med = ops.combined_offdiag_median(F, P, 1.0, 1.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ei, ew = ops.combined_threshold_edges(F, P, float(med), 1.0, 1.0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print("N=%d d=%d: median %.1f ms (one sweep when the sampled bracket holds: %.0f TFLOP/s), edges %.1f ms (2 sweeps, %d edges = %.1f GB written); K itself would be %.1f GB"
      % (N, d, (t1 - t0) * 1e3, 2.0 * N * N * d / (t1 - t0) / 1e12, (t2 - t1) * 1e3, ei.shape[1], ei.shape[1] * 20 / 1e9, N * N * 4 / 1e9))
