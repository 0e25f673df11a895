"""Fast path on clustered data (near-duplicate patches): fallback rate and time vs the spread inside a cluster."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
dev = torch.device('cuda')
N, d, C = 65536, 512, 512
g = torch.Generator(device=dev).manual_seed(7)
centers = torch.randn((C, d), generator=g, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
assign = torch.randint(0, C, (N,), generator=g, device=dev)
for sigma in (0.3, 0.1, 0.03, 0.01, 0.003):
    X = centers[assign] + sigma * torch.randn((N, d), generator=g, device=dev) / d ** 0.5
    X /= X.norm(dim=1, keepdim=True)
    for metric in ("cosine", "neg_sq_l2"):
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            i, v, st = mmf.simtopk(X, metric=metric, k=5, return_stats=True, profile=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        te = None
        if sigma in (0.3, 0.003):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ie, ve = mmf.simtopk(X, metric=metric, k=5, precision="exact")
            torch.cuda.synchronize(); te = (time.perf_counter() - t0) * 1e3
            assert torch.equal(i, ie) and torch.equal(v, ve)
        print("sigma=%.3f %-9s: wall %.2f ms scan %.2f rerank %.2f fallback %.2f ms rows %d (overflow %d) cand/row %.1f%s" % (
            sigma, metric, dt, st['scan_ms'], st['rerank_ms'], st['fallback_ms'], st['fallback_rows'], st['overflow_rows'],
            st['candidates'] / N, "" if te is None else "  | exact path %.2f ms, identical" % te), flush=True)
