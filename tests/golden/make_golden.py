#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE ITSELF.

Run once in the build container (the reference mount /root/reference does not exist on the GPU
box, and nothing at test time reads it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

How the reference code is executed (no reference source is copied into this repo):
  * build_hypergraph/similarity_kernel.py and hypergraph/build_hypergraph/similarity_kernel.py are
    imported from their files as-is (they need only torch/numpy).
  * build_hypergraph/preprocess_hypergraph.py imports h5py at module level, which this image does
    not have and which the arithmetic functions never touch.  Instead of faking that library, the
    FunctionDef nodes of the four arithmetic functions are compiled from the reference file's AST
    and executed in a namespace holding the names they use (torch, numpy, F, sklearn's KMeans and
    NearestNeighbors, and the reference's own compute_combined_similarity).  The h5 I/O functions
    are not run; their contract is documented in DESIGN.md from the source text.

Only arrays (inputs + the reference's outputs) are written.  Versions used are recorded in
meta.json next to the fixtures.

    --only g8,signatures     regenerate just those pieces (the others are left as committed)

G8 is the arithmetic of the two PIPELINES (process_single_file :566-585 and rebuild_hypergraph_from_similarity
:818-897) obtained by calling the reference's own four arithmetic functions in the order and with the arguments those
pipelines use; the HDF5 reads / writes around them cannot run here (h5py is not in the image) and are not faked.
signatures.json additionally records the signatures of the seven HDF5 / pipeline functions: their `def` statements
are compiled from the reference file's AST (annotations only need typing names), never called.
"""
from __future__ import annotations

import ast
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("MMF_REFERENCE_ROOT", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_module(path: str, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_functions(path: str, names, namespace: dict) -> dict:
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in keep}
    if missing:
        raise RuntimeError(f"reference functions not found: {missing}")
    code = compile(ast.Module(body=keep, type_ignores=[]), path, "exec")
    exec(code, namespace)
    return {n: namespace[n] for n in names}


def unit_rows(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)


def positions(n, dp, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, dp, generator=g, dtype=torch.float32)


def main() -> None:
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))

    def want(tag):
        return only is None or tag in only

    def save(name, **arrays):
        if want(name.split("_")[0]):
            np.savez_compressed(os.path.join(OUT, name), **arrays)
    torch.set_num_threads(8)
    sk = load_module(os.path.join(REF, "build_hypergraph", "similarity_kernel.py"), "ref_sk_primary")
    sk2 = load_module(os.path.join(REF, "hypergraph", "build_hypergraph", "similarity_kernel.py"), "ref_sk_second")
    from sklearn.cluster import KMeans
    from sklearn.neighbors import NearestNeighbors
    import sklearn
    from typing import Dict, List, Optional, Tuple
    ns = dict(torch=torch, np=np, F=F, KMeans=KMeans, NearestNeighbors=NearestNeighbors, Dict=Dict,
              List=List, Optional=Optional, Tuple=Tuple,
              compute_combined_similarity=sk.compute_combined_similarity)
    pp = load_functions(os.path.join(REF, "build_hypergraph", "preprocess_hypergraph.py"),
                        ["compute_wsi_tma_similarity", "build_hypergraph_knn_kmeans",
                         "aggregate_wsi_super_patches", "group_by_similarity"], ns)

    # ---- G1: dense a1/a2/a3 ---------------------------------------------------------------------
    g1 = {}
    for (N, D) in [(2, 4), (64, 32), (256, 128)]:
        X = unit_rows(N, D, 1234 + N)
        P2 = positions(N, 2, 99 + N)
        P3 = positions(N, 3, 199 + N)
        g1[f"N{N}_D{D}_X"] = X.numpy()
        g1[f"N{N}_D{D}_P2"] = P2.numpy()
        g1[f"N{N}_D{D}_P3"] = P3.numpy()
        for lam in ((0.5, 1.0, 2.0) if N <= 64 else (1.0,)):
            g1[f"N{N}_D{D}_lam{lam}_Kh"] = sk.compute_morphological_similarity(X, lam).numpy()
            g1[f"N{N}_D{D}_lam{lam}_Kg2"] = sk.compute_spatial_similarity(P2, lam).numpy()
            g1[f"N{N}_D{D}_lam{lam}_Kg3"] = sk.compute_spatial_similarity(P3, lam).numpy()
            g1[f"N{N}_D{D}_lam{lam}_K2"] = sk.compute_combined_similarity(X, P2, lam, lam).numpy()
            g1[f"N{N}_D{D}_lam{lam}_K3"] = sk.compute_combined_similarity(X, P3, lam, 2.0 * lam).numpy()
        Z = torch.zeros(N, 2)
        g1[f"N{N}_D{D}_Kzero"] = sk.compute_combined_similarity(X, Z, 1.0, 1.0).numpy()
        # the second copy of the file must agree bit for bit
        assert torch.equal(sk2.compute_morphological_similarity(X, 1.0), sk.compute_morphological_similarity(X, 1.0))
    save("g1_dense.npz", **g1)

    # ---- G2: a4 / a6 threshold edge builder -----------------------------------------------------
    g2 = {}
    for N in (2, 8, 64):
        X = unit_rows(N, 16, 777 + N)
        P = positions(N, 2, 888 + N)
        g2[f"N{N}_X"] = X.numpy()
        g2[f"N{N}_P"] = P.numpy()
        for ratio in (0.0, 0.5, 1.0, 2.0):
            ei, ew = sk.build_weighted_hypergraph(X, P, 1.0, 1.0, ratio)
            g2[f"N{N}_r{ratio}_ei"] = ei.numpy()
            g2[f"N{N}_r{ratio}_ew"] = ew.numpy()
        d1 = sk.build_hypergraph_data(X, P, 1.0, 1.0, 0.5, True)
        d2 = sk2.build_hypergraph_data(X, P, 1.0, 1.0, 0.5, True)
        assert sorted(d1.keys()) == ["edge_attr", "edge_index", "pooled_feature", "pos", "x"]
        assert sorted(d2.keys()) == ["edge_attr", "edge_index", "pooled_features", "pos", "x"]
        g2[f"N{N}_data_ei"] = d1["edge_index"].numpy()
        g2[f"N{N}_data_ew"] = d1["edge_attr"].numpy()
        g2[f"N{N}_data_pool"] = d1["pooled_feature"].numpy()
        assert torch.equal(d1["pooled_feature"], d2["pooled_features"])
    errs = {}
    try:
        sk.build_weighted_hypergraph(unit_rows(4, 8, 1), positions(4, 2, 2), 1.0, 1.0, None)
        errs["ratio_none"] = "no error"
    except Exception as e:  # noqa: BLE001
        errs["ratio_none"] = type(e).__name__
    try:
        sk.build_weighted_hypergraph(unit_rows(1, 8, 1), positions(1, 2, 2), 1.0, 1.0, 0.5)
        errs["n1"] = "no error"
    except Exception as e:  # noqa: BLE001
        errs["n1"] = type(e).__name__
    g2["errors_json"] = np.array(json.dumps(errs))
    save("g2_threshold.npz", **g2)

    # ---- G3: a7 cross-modal dense ---------------------------------------------------------------
    g3 = {}
    for (N, M, D) in [(8, 12, 32), (100, 37, 128)]:
        A = unit_rows(N, D, 1234 + N)
        B = unit_rows(M, D, 4321 + M)
        S, st = pp["compute_wsi_tma_similarity"](A, positions(N, 2, 5), B, 1.0, 1.0)
        S2, st2 = pp["compute_wsi_tma_similarity"](A, positions(N, 2, 5), B, 0.5, 3.0)
        g3[f"N{N}_M{M}_A"] = A.numpy()
        g3[f"N{N}_M{M}_B"] = B.numpy()
        g3[f"N{N}_M{M}_S_lam1"] = S.numpy()
        g3[f"N{N}_M{M}_S_lam0.5"] = S2.numpy()
        g3[f"N{N}_M{M}_stats_lam1"] = np.array([st[k] for k in ("mean", "std", "min", "max", "median")], dtype=np.float64)
        g3[f"N{N}_M{M}_stats_lam0.5"] = np.array([st2[k] for k in ("mean", "std", "min", "max", "median")], dtype=np.float64)
    save("g3_cross.npz", **g3)

    # ---- G4: a8 sklearn kNN on tie-free data ----------------------------------------------------
    g4 = {}
    for (N, D) in [(64, 32), (512, 128)]:
        X = unit_rows(N, D, 1234 + N + D)
        g4[f"N{N}_D{D}_X"] = X.numpy()
        for k in (1, 5, 16):
            knn = NearestNeighbors(n_neighbors=k + 1, metric="euclidean")   # preprocess_hypergraph.py:379
            knn.fit(X.numpy())
            dist, ind = knn.kneighbors(X.numpy())
            g4[f"N{N}_D{D}_k{k}_ind"] = ind.astype(np.int64)
            g4[f"N{N}_D{D}_k{k}_dist"] = dist.astype(np.float64)
    save("g4_knn.npz", **g4)

    # ---- G5: a8+a9+a10 whole build_hypergraph_knn_kmeans ----------------------------------------
    g5 = {}
    for tag, (Nw, Nt, D, k, H, zero_row) in {"small": (40, 24, 32, 5, 4, False),
                                               "zero": (30, 10, 16, 3, 3, True),
                                               "mid": (100, 60, 128, 5, 10, False)}.items():
        W = unit_rows(Nw, D, 31 + Nw)
        T = unit_rows(Nt, D, 41 + Nt)
        if zero_row:
            W[3] = 0.0
        ei, ew, st = pp["build_hypergraph_knn_kmeans"](W, T, np.zeros(Nw, dtype=np.int64), k, H)
        allf = torch.cat([W, T], 0).numpy()
        labels = KMeans(n_clusters=H, random_state=42, n_init=10).fit_predict(allf)   # :391-392
        e = ei.numpy()
        order = np.lexsort((e[1], e[0]))
        g5[f"{tag}_W"] = W.numpy()
        g5[f"{tag}_T"] = T.numpy()
        g5[f"{tag}_k"] = np.array(k)
        g5[f"{tag}_H"] = np.array(H)
        g5[f"{tag}_labels"] = labels.astype(np.int64)
        g5[f"{tag}_ei_sorted"] = e[:, order]
        g5[f"{tag}_ew_sorted"] = ew.numpy()[order]
        g5[f"{tag}_num_edges"] = np.array(st["num_edges"])
    save("g5_knn_kmeans.npz", **g5)

    # ---- G6: tie cases --------------------------------------------------------------------------
    g6 = {}
    Xd = unit_rows(32, 16, 5)
    Xd[7] = Xd[3]
    Xd[20] = Xd[3]                       # duplicate rows: self is not at column 0 for all of them
    lat = torch.tensor([[float(i), float(j)] for i in range(6) for j in range(6)])   # integer lattice
    for tag, X in (("dup", Xd), ("lattice", lat)):
        knn = NearestNeighbors(n_neighbors=6, metric="euclidean").fit(X.numpy())
        dist, ind = knn.kneighbors(X.numpy())
        g6[f"{tag}_X"] = X.numpy()
        g6[f"{tag}_ind"] = ind.astype(np.int64)
        g6[f"{tag}_dist"] = dist.astype(np.float64)
    save("g6_ties.npz", **g6)

    # ---- G7: a5 both signatures -----------------------------------------------------------------
    X = unit_rows(50, 24, 9)
    P = positions(50, 2, 10)
    save("g7_pool.npz", X=X.numpy(), P=P.numpy(),
                        pool1=sk.mean_pool_with_similarity(X).numpy(),
                        pool2=sk2.mean_pool_with_similarity(X, P, 1.0, 1.0).numpy())

    # ---- G8: the two pipelines' arithmetic, composed from the reference's own functions -----------
    g8 = {}
    Nw, Nt, D, S, G, k, H = 600, 48, 64, 40, 6, 5, 8
    gen = torch.Generator().manual_seed(2024)
    cent = torch.randn(60, D, generator=gen)
    Wf = cent[torch.randint(0, 60, (Nw,), generator=gen)] * 0.4 + 0.05 * torch.randn(Nw, D, generator=gen)   # clustered patches
    Wp = torch.rand(Nw, 2, generator=gen)
    Tf = cent[torch.randint(0, 60, (Nt,), generator=gen)] * 0.4 + 0.05 * torch.randn(Nt, D, generator=gen)
    lam_h, lam_g = 0.5, 2.0
    # process_single_file, preprocess_hypergraph.py:566-585
    sf, sp, wst, Kw = pp["aggregate_wsi_super_patches"](Wf, Wp, S, lam_h, lam_g, torch.device("cpu"))
    sim, sst = pp["compute_wsi_tma_similarity"](sf, sp, Tf, lam_h, lam_g, torch.device("cpu"))
    gl, gst = pp["group_by_similarity"](sim, G, method="kmeans")
    ei, ew, hst = pp["build_hypergraph_knn_kmeans"](sf, Tf, gl, k, H, torch.device("cpu"))
    e = ei.numpy()
    order = np.lexsort((e[1], e[0]))
    g8.update(wsi_features=Wf.numpy(), wsi_positions=Wp.numpy(), tma_features=Tf.numpy(),
              params=np.array([S, G, k, H]), lambdas=np.array([lam_h, lam_g]),
              super_features=sf.numpy(), super_positions=sp.numpy(), K_wsi=Kw.numpy(), sim=sim.numpy(),
              group_labels=np.asarray(gl).astype(np.int64), ei_sorted=e[:, order], ew_sorted=ew.numpy()[order],
              wsi_stats=np.array([wst["avg_intra_cluster_similarity"]] + [wst["wsi_similarity_matrix_stats"][q] for q in ("mean", "std", "min", "max", "median")], dtype=np.float64),
              sim_stats=np.array([sst[q] for q in ("mean", "std", "min", "max", "median")], dtype=np.float64),
              group_sizes=np.array(gst["group_sizes"], dtype=np.int64), num_edges=np.array(hst["num_edges"]))
    # rebuild_hypergraph_from_similarity with num_wsi_super_patches = S2, num_groups = G2 (:818-867: lambdas are the
    # literal 1.0s of the source, the stored K_wsi is handed to the aggregation), then the edge-weight median
    # filter of :885-897 (three torch lines restated here on the reference's own edge weights)
    S2, G2, ratio = 25, 4, 1.0
    sf2, sp2, wst2, _ = pp["aggregate_wsi_super_patches"](Wf, Wp, S2, lambda_h=1.0, lambda_g=1.0, device=torch.device("cpu"),
                                                          wsi_similarity_matrix=Kw)
    sim2, sst2 = pp["compute_wsi_tma_similarity"](sf2, sp2, Tf, lambda_h=1.0, lambda_g=1.0, device=torch.device("cpu"))
    gl2, gst2 = pp["group_by_similarity"](sim2, G2, method="kmeans")
    ei2, ew2, hst2 = pp["build_hypergraph_knn_kmeans"](sf2, Tf, gl2, k, H, torch.device("cpu"))
    med = ew2.median().item()
    thr = med * ratio
    mask = ew2 >= thr
    e2 = ei2[:, mask].numpy()
    w2 = ew2[mask].numpy()
    order2 = np.lexsort((e2[1], e2[0]))
    g8.update(rb_params=np.array([S2, G2]), rb_ratio=np.array(ratio), rb_super_features=sf2.numpy(), rb_super_positions=sp2.numpy(),
              rb_sim=sim2.numpy(), rb_group_labels=np.asarray(gl2).astype(np.int64), rb_num_edges=np.array(hst2["num_edges"]),
              rb_median=np.array(med, dtype=np.float64), rb_threshold=np.array(thr, dtype=np.float64),
              rb_ei_sorted=e2[:, order2], rb_ew_sorted=w2[order2],
              rb_sim_stats=np.array([sst2[q] for q in ("mean", "std", "min", "max", "median")], dtype=np.float64))
    save("g8_pipeline.npz", **g8)

    # ---- signatures of the boundary functions (names, order, defaults): data, not code ----------
    import inspect

    def sig(fn):
        return [[p.name, None if p.default is inspect._empty else repr(p.default)] for p in inspect.signature(fn).parameters.values()]
    sigs = {"build_hypergraph": {n: sig(getattr(sk, n)) for n in
                                 ("compute_morphological_similarity", "compute_spatial_similarity", "compute_combined_similarity",
                                  "build_weighted_hypergraph", "mean_pool_with_similarity", "build_hypergraph_data")},
            "hypergraph.build_hypergraph": {n: sig(getattr(sk2, n)) for n in
                                            ("compute_morphological_similarity", "compute_spatial_similarity",
                                             "compute_combined_similarity", "build_weighted_hypergraph",
                                             "mean_pool_with_similarity", "build_hypergraph_data")}}
    sigs["build_hypergraph"].update({n: sig(pp[n]) for n in pp})
    # the HDF5 / pipeline functions: `def` statements only (never called — h5py is absent)
    io_names = ["load_wsi_data", "load_tma_data", "save_hypergraph_to_h5", "process_single_file", "process_dataset",
                "load_similarity_matrices", "rebuild_hypergraph_from_similarity", "batch_rebuild_hypergraph"]
    io = load_functions(os.path.join(REF, "build_hypergraph", "preprocess_hypergraph.py"), io_names, dict(ns))
    sigs["build_hypergraph"].update({n: sig(io[n]) for n in io_names})
    ref_init = ast.parse(open(os.path.join(REF, "build_hypergraph", "__init__.py"), encoding="utf-8").read())
    ref_all = [ast.literal_eval(n.value) for n in ref_init.body
               if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "__all__"][0]
    sigs["build_hypergraph.__all__"] = list(ref_all)
    if want("signatures"):
        with open(os.path.join(OUT, "signatures.json"), "w") as fh:
            json.dump(sigs, fh, indent=1, sort_keys=True)

    meta = {"generator": "tests/golden/make_golden.py", "reference": "zz9tf/multimodal-fusion @ 2026-01-30",
            "torch": torch.__version__, "numpy": np.__version__, "sklearn": sklearn.__version__,
            "python": sys.version.split()[0], "torch_threads": torch.get_num_threads()}
    if only is None:
        with open(os.path.join(OUT, "meta.json"), "w") as fh:
            json.dump(meta, fh, indent=1)
    print("wrote fixtures to", OUT, meta)


if __name__ == "__main__":
    main()
