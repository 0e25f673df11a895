"""One-off stress of the dense side against the CPU oracle / torch: sim_dense (all metrics, dtypes, ragged shapes),
sim_dense_combined, the streaming median / edge builder against the materialised one, medians and statistics around the
one-sweep threshold, threshold_edges.  scripts/stress_dense.py [cases] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
import oracle
ops = mmf.ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad, t0 = 0, time.time()
def fail(*a):
    global bad
    bad += 1
    print("MISMATCH", *a, flush=True)
for case in range(cases):
    g = torch.Generator(device="cuda").manual_seed(9000 + case)
    n, m = int(rng.randint(1, 600)), int(rng.randint(1, 900))
    d = int(rng.choice([1, 2, 3, 8, 9, 16, 31, 32, 33, 64, 100, 257, 512, 640, 1000]))
    metric = str(rng.choice(["dot", "cosine", "neg_sq_l2", "rbf", "rbf_direct"]))
    dt = [torch.float32, torch.float16, torch.bfloat16][int(rng.randint(0, 3))]
    X = (torch.randn((n, d), generator=g, device="cuda") * 0.2).to(dt)
    Y = (torch.randn((m, d), generator=g, device="cuda") * 0.2).to(dt)
    got = ops.sim_dense(X, Y, metric=metric, lam=0.7).cpu().numpy()
    ref = oracle.sim_dense(X.float().cpu().numpy(), Y.float().cpu().numpy(), metric=metric, lam=0.7)
    if metric in ("dot", "cosine", "neg_sq_l2"):
        if not np.array_equal(got, ref): fail("sim_dense", case, n, m, d, metric, dt)
    elif not np.allclose(got, ref, rtol=0, atol=1e-5): fail("sim_dense", case, n, m, d, metric, dt)
    if case % 3 == 0:                                   # combined similarity, streaming == materialised, edges == oracle
        N = int(rng.randint(2, 700)); D = int(rng.choice([3, 16, 64, 100, 512])); dp = int(rng.choice([2, 3]))
        F = torch.randn((N, D), generator=g, device="cuda") * 0.1
        P = torch.rand((N, dp), generator=g, device="cuda") * 2
        K = ops.sim_dense_combined(F, P, 0.7, 0.3)
        refK = oracle.sim_dense_combined(F.cpu().numpy(), P.cpu().numpy(), 0.7, 0.3)
        if not np.allclose(K.cpu().numpy(), refK, rtol=0, atol=1e-5): fail("combined", case, N, D, dp)
        med = ops.offdiag_lower_median(K)
        if float(med) != oracle.offdiag_lower_median(K.cpu().numpy()): fail("median", case, N)
        pr = int(rng.choice([0, 64, 128, 200]))
        if not torch.equal(med, ops.combined_offdiag_median(F, P, 0.7, 0.3, pr)): fail("stream median", case, N, pr)
        thr = float(med) * float(rng.choice([0.8, 1.0, 1.1]))
        ei, ew = ops.threshold_edges(K, thr)
        oi, ow = oracle.threshold_edges(K.cpu().numpy(), np.float32(thr))
        if not (np.array_equal(ei.cpu().numpy(), oi) and np.array_equal(ew.cpu().numpy(), ow)): fail("edges", case, N)
        si, sw = ops.combined_threshold_edges(F, P, thr, 0.7, 0.3, pr)
        if not (torch.equal(si, ei) and torch.equal(sw, ew)): fail("stream edges", case, N, pr)
    if case % 5 == 0:                                   # order statistics around the one-sweep threshold (2^22 values)
        cnt = int(rng.choice([4194303, 4194304, 4194305, 5000011, 3999999]))
        v = torch.rand(cnt, generator=g, device="cuda")
        mode = int(rng.randint(0, 3))
        if mode == 1: v = torch.round(v * 7) / 7        # heavy ties: the bracket buffer overflows, radix fallback
        if mode == 2: v = 0.5 + 1e-5 * (v - 0.5)
        if float(ops.lower_median(v)) != float(v.median()): fail("lower_median", case, cnt, mode)
        st = ops.array_stats(v)
        if st["median"] != float(v.median()) or st["min"] != float(v.min()) or st["max"] != float(v.max()): fail("array_stats", case, cnt, mode)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})", cases, "cases")
