import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
N, d, P, r = 1048576, 1024, 8, 5
X = torch.empty((N, d), dtype=torch.float16, device='cuda')
for b in range(0, N, 65536):
    g = torch.Generator(device='cuda').manual_seed(5000 + b)
    blk = torch.randn((65536, d), generator=g, device='cuda', dtype=torch.float32)
    X[b:b + 65536] = (blk / blk.norm(dim=1, keepdim=True)).half()
lo, hi = r * (N // P), (r + 1) * (N // P)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, val, st = mmf.simtopk(X[lo:hi], X, metric='cosine', k=5, exclude_self=True, row_offset=lo, return_stats=True, profile=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
fl = 2.0 * (hi - lo) * N * d
print("C5 one rank of 8: wall %.1f ms scan %.1f ms (%.0f TFLOP/s, %.1f %% of peak) prep %.2f rerank %.2f fallback rows %d" % (dt, st['scan_ms'], fl / st['scan_ms'] / 1e9, fl / st['scan_ms'] / 1e9 / 25, st['prep_ms'], st['rerank_ms'], st['fallback_rows']))
