"""Golden G9: scikit-learn's own KMeans labels at the pipeline's real shape (N = 16384, d = 512, k = 100 super-patches,
build_hypergraph/preprocess_hypergraph.py:150-151 with num_super_patches = 100, :516), produced in the build container
(8 cores, numpy's bundled OpenBLAS), together with the labels, seeds and iteration counts of the CPU restatement
(oracle/kmeans_restate.py) on the same data.  The data are regenerated from their seeds by tests / scripts
(numpy default_rng is reproducible across machines); only the labels are stored.

    python tests/golden/make_g9_kmeans.py        # ~5 minutes, CPU only
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sklearn
from sklearn.cluster import KMeans
from oracle import kmeans_restate as kr


def g9_data(kind: str, n: int = 16384, d: int = 512) -> np.ndarray:
    rng = np.random.default_rng(20260 + (0 if kind == "clustered" else 1))
    if kind == "clustered":      # unit-norm rows around 128 directions: what patch embeddings of one slide look like
        c = rng.standard_normal((128, d)).astype(np.float32)
        X = (c[rng.integers(0, 128, n)] + 0.7 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
        return X / np.linalg.norm(X, axis=1, keepdims=True)
    return rng.standard_normal((n, d)).astype(np.float32)      # no cluster structure at all: the least stable case


if __name__ == "__main__":
    out = {}
    for kind in ("clustered", "gauss"):
        X = g9_data(kind)
        t0 = time.time()
        km = KMeans(n_clusters=100, random_state=42, n_init=10).fit(X)
        t1 = time.time()
        info = {}
        rl = kr.kmeans_fit_predict(X, 100, info=info)
        t2 = time.time()
        out[f"{kind}_sklearn_labels"] = km.labels_.astype(np.int16)
        out[f"{kind}_sklearn_inertia"] = np.float64(km.inertia_)
        out[f"{kind}_restate_labels"] = rl.astype(np.int16)
        out[f"{kind}_restate_inertia"] = np.float64(info["inertia"])
        out[f"{kind}_restate_seeds"] = np.stack([p["seeds"] for p in info["per_init"]]).astype(np.int32)
        out[f"{kind}_restate_n_iter"] = np.array([p["n_iter"] for p in info["per_init"]], np.int32)
        out[f"{kind}_restate_best"] = np.int32(info["best_init"])
        print(kind, "sklearn == restate:", np.array_equal(km.labels_, rl), "mismatched", int((km.labels_ != rl).sum()),
              "inertia", km.inertia_, info["inertia"], "ambiguous", info["ambiguous"], f"sklearn {t1 - t0:.0f}s restate {t2 - t1:.0f}s", flush=True)
    out["versions"] = np.array([f"sklearn {sklearn.__version__}", f"numpy {np.__version__}", f"cpus {os.cpu_count()}"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g9_kmeans_scale.npz"), **out)
