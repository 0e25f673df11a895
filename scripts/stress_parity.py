"""One-off stress: many seeded random configurations of mmf_simtopk against the CPU oracle (the shapes of
tests/test_gpu_properties.py::test_random_configurations_against_the_oracle, with k up to 43, all precisions, clustered
rows, the scan's query order forced on in half of the cases, more cases).  scripts/stress_parity.py [cases] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
import oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
metrics = ["cosine", "dot", "neg_sq_l2", "rbf"]
dtypes = [torch.float32, torch.float16, torch.bfloat16]
bad, t0 = 0, time.time()
for case in range(cases):
    n = int(rng.randint(1, 900))
    m = int(rng.randint(45, 4000))
    d = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 100, 129, 255, 256, 300, 512, 513, 777, 1024, 1100, 1536]))
    k = int(rng.choice([1, 3, 5, 8, 11, 16, 19, 20, 21, 27, 32, 40, 43, 44, 45, 64, 90, 130]))
    metric = metrics[int(rng.randint(0, 4))]
    dt = dtypes[int(rng.randint(0, 3))]
    excl = bool(rng.randint(0, 2))
    ro, co = int(rng.randint(0, 50)), int(rng.randint(0, 50))
    splits = int(rng.choice([0, 0, 1, 2, 8]))
    prec = str(rng.choice(["auto", "auto", "exact", "fast", "fast_bf16"]))
    order = str(rng.choice(["on", "off"]))
    k = min(k, m - 1)
    if prec in ("fast", "fast_bf16") and not mmf.ops.fast_scan_supported(d, k, excl):
        prec = "auto"
    scale = 0.05 if metric == "rbf" else 1.0
    g = torch.Generator(device="cuda").manual_seed(5000 + case)
    kind = int(rng.randint(0, 4))
    X = torch.randn((n, d), generator=g, device="cuda")
    Y = torch.randn((m, d), generator=g, device="cuda")
    if kind == 1:                                           # near-duplicate candidates: crowded lists, overflow lists
        c = torch.randn((max(2, m // 60), d), generator=g, device="cuda")
        Y = c[torch.randint(0, c.shape[0], (m,), generator=g, device="cuda")] + 1e-3 * Y
        X = c[torch.randint(0, c.shape[0], (n,), generator=g, device="cuda")] + 1e-3 * X
    X, Y = (X * scale).to(dt), (Y * scale).to(dt)
    if kind == 2:
        Y[: min(n, m)] = X[: min(n, m)]                     # exact ties, self columns
    try:
        idx, val = mmf.simtopk(X, Y, metric=metric, lam=0.7, k=k, exclude_self=excl, row_offset=ro, col_offset=co,
                               col_splits=splits, precision=prec, query_order=order)
    except RuntimeError as e:
        print("ERROR", case, n, m, d, k, metric, dt, excl, ro, co, splits, prec, order, kind, str(e)[:120], flush=True)
        bad += 1
        continue
    ri, rv = oracle.simtopk(X.float().cpu().numpy(), Y.float().cpu().numpy(), metric=metric, lam=0.7, k=k,
                            exclude_self=excl, row_offset=ro, col_offset=co)
    ok = np.array_equal(idx.cpu().numpy(), ri) and (np.allclose(val.cpu().numpy(), rv, rtol=0, atol=1e-5) if metric == "rbf"
                                                     else np.array_equal(val.cpu().numpy(), rv))
    if not ok:
        bad += 1
        print("MISMATCH", case, n, m, d, k, metric, dt, excl, ro, co, splits, prec, order, kind, flush=True)
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})", cases, "cases")
