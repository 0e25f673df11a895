"""How often does the decision-level restatement of scikit-learn's KMeans (oracle/kmeans_restate.py) return
scikit-learn's labels?  CPU only.  python scripts/kmeans_restate_vs_sklearn.py [cases]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.cluster import KMeans
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
    if kind == "sim":      # rows of an RBF similarity matrix (group_by_similarity)
        a = rng.standard_normal((n, 16)).astype(np.float32); b = rng.standard_normal((d, 16)).astype(np.float32)
        return np.exp(-0.05 * ((a[:, None, :] - b[None]) ** 2).sum(-1)).astype(np.float32)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(7)
ok = 0
for t in range(cases):
    kind = ["gauss", "blobs", "unit", "sim"][t % 4]
    n = int(rng.choice([40, 64, 160, 600, 1500, 4000]))
    d = int(rng.choice([16, 32, 64, 128]))
    k = int(rng.choice([3, 6, 10, 25, 40]))
    k = min(k, n // 4)
    X = data(kind, n, d, rng)
    t0 = time.time(); ref = KMeans(n_clusters=k, random_state=42, n_init=10).fit_predict(X); t1 = time.time()
    info = {}
    lab = kr.kmeans_fit_predict(X, k, info=info); t2 = time.time()
    same = np.array_equal(ref, lab)
    ok += same
    print(f"{kind:6s} n={n:5d} d={d:4d} k={k:3d} same={same} mismatched={int((ref != lab).sum()):5d} amb={info['ambiguous']} "
          f"best_init={info['best_init']} sk={t1 - t0:.2f}s re={t2 - t1:.2f}s", flush=True)
print(f"{ok}/{cases} identical")
