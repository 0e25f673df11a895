"""threshold_edges: what bounds the fill pass?  Same matrix, thresholds that keep nothing / half / everything."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
ops = mmf.ops
dev = torch.device('cuda')
N = 16384
g = torch.Generator(device=dev).manual_seed(1)
K = torch.rand((N, N), generator=g, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for thr in (2.0, 0.75, 0.5, 0.25, -1.0):
    ei, ew = ops.threshold_edges(K, thr)
    E = ei.shape[1]
    t = timed(lambda: ops.threshold_edges(K, thr))
    print(f"thr={thr:5.2f} edges={E:10d} {t:7.3f} ms  read {2 * N * N * 4 / t / 1e6:7.1f} GB/s  total {(2 * N * N * 4 + E * 20) / t / 1e6:7.1f} GB/s", flush=True)
t = timed(lambda: ops.offdiag_lower_median(K))
print(f"offdiag_lower_median {t:7.3f} ms  {4 * N * N * 4 / t / 1e6:7.1f} GB/s (4 passes)")
