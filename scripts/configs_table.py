"""Timings of BASELINE.json's single-GPU configurations (C2, C3) and of the headline workload (C4 on one GPU)."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device('cuda')


def run(name, X, Y, metric, **kw):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i, v, st = mmf.simtopk(X, Y, metric=metric, k=5, return_stats=True, profile=True, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    n, m, d = X.shape[0], (Y if Y is not None else X).shape[0], X.shape[1]
    print("%-44s %7.2f ms  %.3e pairs/s  scan %.2f ms = %4.0f TFLOP/s (%.1f %%)  rescans %d" % (
        name, dt, n * m / dt * 1e3, st['scan_ms'], 2.0 * n * m * d / st['scan_ms'] / 1e9, 2.0 * n * m * d / st['scan_ms'] / 1e9 / 25, st['fallback_rows']), flush=True)


X = make_rows(0, 65536, 512, dev)
g = torch.Generator(device=dev).manual_seed(4321)
Y = torch.randn((65536, 512), generator=g, device=dev); Y /= Y.norm(dim=1, keepdim=True)
run("C2 N=65536 d=512 cosine", X, None, "cosine")
run("C2 N=65536 d=512 rbf (lambda 1)", X, None, "rbf", lam=1.0)
run("C2 N=65536 d=512 neg_sq_l2 (sklearn k-NN)", X, None, "neg_sq_l2")
run("C3 N=M=65536 d=512 two modalities, cosine", X, Y, "cosine", exclude_self=False)
X4 = make_rows(0, 262144, 512, dev)
run("C4 N=262144 d=512 cosine (one GPU)", X4, None, "cosine")
