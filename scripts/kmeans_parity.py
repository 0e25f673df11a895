"""Device KMeans (mmf_kmeans_fit) against scikit-learn's own labels and against the CPU restatement
(oracle/kmeans_restate.py), restart by restart.  GPU box:  python scripts/kmeans_parity.py [--big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sklearn.cluster import KMeans
import multimodal_fusion_amd as mmf
from importlib import import_module
km = import_module("multimodal_fusion_amd.kmeans")
from oracle import kmeans_restate as kr


def data(kind, n, d, rng):
    if kind == "gauss":
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == "blobs":
        c = rng.standard_normal((max(2, n // 40), d)).astype(np.float32) * 2
        return (c[rng.integers(0, len(c), n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    if kind == "unit":
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x / np.linalg.norm(x, axis=1, keepdims=True)
    a = rng.standard_normal((n, 16)).astype(np.float32); b = rng.standard_normal((d, 16)).astype(np.float32)
    return np.exp(-0.05 * ((a[:, None, :] - b[None]) ** 2).sum(-1)).astype(np.float32)


def one(X, k, tag, restate=True):
    n = X.shape[0]
    t0 = time.time(); ref = KMeans(n_clusters=k, random_state=42, n_init=10).fit(X); t1 = time.time()
    Xg = torch.from_numpy(X).cuda()
    first, u = km.sklearn_stream(42, 10, k, n)
    lab, C, info, seeds = mmf.ops.kmeans_fit(Xg, k, first, u, return_seeds=True)
    torch.cuda.synchronize(); t2 = time.time()
    lab2, _, info2 = mmf.ops.kmeans_fit(Xg, k, first, u)
    torch.cuda.synchronize(); t3 = time.time()
    lab = lab.cpu().numpy()
    same_sk = np.array_equal(lab, ref.labels_)
    msg = f"{tag:28s} vs sklearn: same={same_sk} mismatched={int((lab != ref.labels_).sum())} inertia {info['inertia']:.6g} / {ref.inertia_:.6g} " \
          f"best={info['best_init']} iters={info['n_iter_per_init']} amb={info['ambiguous_draws']},{info['ambiguous_trials']} " \
          f"gpu {1e3 * (t3 - t2):.1f} ms sk {t1 - t0:.2f} s det={bool(torch.equal(lab2.cpu(), torch.from_numpy(lab)))}"
    ok = same_sk
    if restate:
        ri = {}
        rl = kr.kmeans_fit_predict(X, k, info=ri)
        s_same = all(np.array_equal(seeds[i].cpu().numpy(), ri["per_init"][i]["seeds"]) for i in range(10))
        it_same = [p["n_iter"] for p in ri["per_init"]] == info["n_iter_per_init"]
        msg += f" | vs restate: labels={np.array_equal(lab, rl)} seeds={s_same} iters={it_same} best={ri['best_init'] == info['best_init']}"
        ok = ok and np.array_equal(lab, rl)
        cerr = np.abs(C.cpu().numpy() - ref.cluster_centers_).max()
        msg += f" centres maxerr {cerr:.2e}"
    print(msg, flush=True)
    return ok


rng = np.random.default_rng(7)
good = total = 0
g5 = np.load("tests/golden/g5_knn_kmeans.npz")
for tag in ("small", "zero", "mid"):
    X = np.concatenate([g5[f"{tag}_W"], g5[f"{tag}_T"]], 0)
    good += one(X, int(g5[f"{tag}_H"]), f"g5 {tag} n={len(X)}"); total += 1
for t in range(24):
    kind = ["gauss", "blobs", "unit", "sim"][t % 4]
    n = int(rng.choice([40, 64, 160, 600, 1500, 4000])); d = int(rng.choice([16, 33, 64, 128])); k = int(rng.choice([1, 3, 6, 10, 25, 40]))
    k = max(1, min(k, n // 4))
    good += one(data(kind, n, d, rng), k, f"{kind} n={n} d={d} k={k}"); total += 1
if "--big" in sys.argv:      # golden g9's data: is scikit-learn on THIS machine the scikit-learn of the build container?
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    from make_g9_kmeans import g9_data
    g9 = np.load("tests/golden/g9_kmeans_scale.npz")
    for kind in ("clustered", "gauss"):
        X = g9_data(kind)
        good += one(X, 100, f"g9 {kind} n=16384 d=512 k=100", restate=False); total += 1
        here = KMeans(n_clusters=100, random_state=42, n_init=10).fit(X)
        dev = mmf.ops.kmeans_fit(torch.from_numpy(X).cuda(), 100, *km.sklearn_stream(42, 10, 100, 16384))[0].cpu().numpy()
        sk = g9[f"{kind}_sklearn_labels"]
        print(f"   g9 {kind}: scikit-learn(this host, {os.cpu_count()} cpus) == scikit-learn(build container): {np.array_equal(here.labels_, sk)} "
              f"(inertia {here.inertia_:.7g} vs {float(g9[kind + '_sklearn_inertia']):.7g}); device == build container's: {np.array_equal(dev, sk)}", flush=True)
print(f"{good}/{total} identical")
