"""Large k (k + self in 21..44) on the 16-bit scan's 32-entry lists against the exact f32 scan: same result, time each."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
dev = torch.device('cuda')


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for n, d in ((65536, 512), (65536, 256), (262144, 512)):
    g = torch.Generator(device=dev).manual_seed(n + d)
    X = torch.randn((n, d), generator=g, device=dev)
    X = X / X.norm(dim=1, keepdim=True)
    for k in (16, 20, 24, 32, 43):
        if n > 65536 and k not in (20, 32):
            continue
        fi, fv, st = mmf.simtopk(X, metric="cosine", k=k, precision="fast", return_stats=True)
        tf = timed(lambda: mmf.simtopk(X, metric="cosine", k=k, precision="fast"))
        line = f"N={n} d={d} k={k}: fast {tf:8.2f} ms  cand/row {st['candidates'] / n:6.1f} fallback {st['fallback_rows']}"
        if n <= 65536:
            ei, ev = mmf.simtopk(X, metric="cosine", k=k, precision="exact")
            te = timed(lambda: mmf.simtopk(X, metric="cosine", k=k, precision="exact"))
            same = bool((fi == ei).all()) and bool((fv == ev).all())
            line += f"  exact {te:8.2f} ms  same={same}"
        print(line, flush=True)
