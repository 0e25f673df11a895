"""Query order of the 16-bit scan (csrc/mmf_order.hip): results with the order on / off / auto must be the same bits; timings of
both on the benchmark's two workloads."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_fusion_amd as mmf
import bench
dev = torch.device("cuda", 0)


def same(a, b):
    return bool(torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)))


bad = 0
g = torch.Generator(device=dev).manual_seed(5)
for (n, m, d, k, metric, dt, kind) in [(1000, None, 64, 5, "cosine", torch.float32, "dup"), (4097, None, 128, 9, "neg_sq_l2", torch.float32, "dup"),
                                       (3000, 5000, 200, 5, "dot", torch.float32, "gauss"), (5000, None, 512, 16, "rbf", torch.bfloat16, "dup"),
                                       (2048, 777, 96, 3, "cosine", torch.float16, "dup"), (70000, None, 64, 5, "cosine", torch.float32, "dup"),
                                       (40000, 33000, 128, 7, "neg_sq_l2", torch.float32, "dup"), (33, None, 1000, 4, "cosine", torch.float32, "gauss")]:
    mm = m or n
    if kind == "dup":
        c = torch.randn((max(4, mm // 50), d), generator=g, device=dev)
        Y = (c[torch.randint(0, c.shape[0], (mm,), generator=g, device=dev)] + 0.02 * torch.randn((mm, d), generator=g, device=dev)).to(dt)
    else:
        Y = torch.randn((mm, d), generator=g, device=dev).to(dt)
    X = Y if m is None else (Y[torch.randint(0, mm, (n,), generator=g, device=dev)].float() + 0.01 * torch.randn((n, d), generator=g, device=dev)).to(dt)
    kw = dict(metric=metric, lam=0.05, k=k, precision="fast", return_stats=True)
    args = (X,) if m is None else (X, Y)
    r_off = mmf.simtopk(*args, query_order="off", **kw)
    r_on = mmf.simtopk(*args, query_order="on", **kw)
    r_auto = mmf.simtopk(*args, query_order="auto", **kw)
    ok = same(r_off, r_on) and same(r_off, r_auto)
    bad += not ok
    print(f"n {n:6d} m {mm:6d} d {d:4d} k {k:2d} {metric:9s} {str(dt)[6:]:8s} {kind:5s}: on == off == auto {ok}   near rows {r_on[2]['near_rows']:6d}  "
          f"auto ordered {r_auto[2]['query_order']}  flagged off/on {r_off[2]['fallback_rows']}/{r_on[2]['fallback_rows']}", flush=True)

N, d = 262144, 512
for data in ("gaussian", "clustered"):
    X = bench.make_rows(0, N, d, dev, data=data)
    res = {}
    for mode in ("off", "on", "auto"):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = mmf.simtopk(X, metric="cosine", k=5, precision="fast", return_stats=True, profile=True, query_order=mode)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        res[mode] = r
        st = r[2]
        print(f"{data:9s} order {mode:4s}: call {1e3 * (t1 - t0):7.2f} ms  scan {st['scan_ms']:6.2f}  order {st['order_ms']:5.2f}  re-rank {st['rerank_ms']:5.2f}  "
              f"near rows {st['near_rows']}  ordered {st['query_order']}  flagged {st['fallback_rows']}", flush=True)
    ok = same(res["off"], res["on"]) and same(res["off"], res["auto"])
    bad += not ok
    print(f"{data}: same bits {ok}", flush=True)
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)
