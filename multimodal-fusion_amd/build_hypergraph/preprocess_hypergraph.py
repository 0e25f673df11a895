"""Mirror of the reference's build_hypergraph/preprocess_hypergraph.py: same names, argument order, defaults, return
shapes and error behaviour; the arithmetic runs in the gfx950 kernels behind include/mmf_hg.h.

    reference                                  here
    load_wsi_data / load_tma_data    :31-84    h5io (h5py File protocol, imported when a file is opened)
    aggregate_wsi_super_patches      :87-199   mmf_sim_dense_combined, KMeans, mmf_segment_sort + mmf_segment_mean
                                               (pooling :157-170), mmf_segment_offdiag_mean (:175-184), mmf_array_stats
    compute_wsi_tma_similarity       :202-267  mmf_sim_dense_stats(MMF_RBF_DIRECT): matrix + mean/std/min/max/median
    group_by_similarity              :270-332  KMeans over the rows of the similarity matrix
    build_hypergraph_knn_kmeans      :335-433  mmf_simtopk(MMF_NEG_SQ_L2) for the k-NN (:379-388), mmf_clique_pairs for the
                                               KMeans cliques (:395-400), mmf_knn_pairs for the undirected dedup (:403-404),
                                               mmf_edge_cosine for the weights (:414-420)
    save_hypergraph_to_h5            :436-511  h5io layout writer
    process_single_file / _dataset   :514-678  the same flow
    load_similarity_matrices         :726-755
    rebuild_hypergraph_from_similarity :758-916  incl. the edge-weight median filter (:885-897) on mmf_lower_median
    batch_rebuild_hypergraph         :919-990

KMeans: `sklearn.cluster.KMeans(n_clusters, random_state=42, n_init=10)` in the reference (:150, :299, :391).  The default
backend is the device KMeans of multimodal-fusion_amd/kmeans.py (csrc/mmf_kmeans.hip), which takes scikit-learn's
decisions one by one — its random stream is drawn with numpy's RandomState(42) in scikit-learn's order, every sum is formed
in float64 — and returns scikit-learn's labels (tests/golden g5, g8, g9; scikit-learn's own float32 BLAS sums make its
result machine dependent where a decision hangs on rounding noise: DESIGN.md §4.5).  `set_kmeans_backend("sklearn")`,
MMF_KMEANS_BACKEND=sklearn or --kmeans_backend sklearn runs the reference's own call on the host instead.

Documented divergences: self is dropped from the k-NN by identity instead of "column 0" (A5);
edges come out lexicographically sorted instead of in Python set order (A6); stats hold Python scalars (A3); an
existing hypergraph/ group is overwritten instead of raising (A4); method='knn' of group_by_similarity (A2, broken
upstream) is not provided.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import ops
from . import h5io
from ._common import compute_device, result_device_like_preprocess, to_gpu
from .similarity_kernel import compute_combined_similarity

KMEANS_BACKEND = os.environ.get("MMF_KMEANS_BACKEND", "device")
# "device" : multimodal-fusion_amd/kmeans.py (scikit-learn's fit, decision for decision, on the GPU; labels identical)
# "sklearn": the reference's own call on the host


def set_kmeans_backend(name: str) -> None:
    global KMEANS_BACKEND
    if name not in ("sklearn", "device"):
        raise ValueError("kmeans backend must be 'sklearn' or 'device'")
    KMEANS_BACKEND = name


def _kmeans_labels(x: torch.Tensor, n_clusters: int) -> torch.Tensor:
    """int64 labels on x's (ROCm) device: KMeans(n_clusters, random_state=42, n_init=10).fit_predict(x)."""
    if KMEANS_BACKEND == "device":
        from ..kmeans import kmeans_fit_predict
        return kmeans_fit_predict(x, n_clusters, n_init=10, seed=42)[0]
    if KMEANS_BACKEND != "sklearn":
        raise ValueError(f"unknown KMeans backend {KMEANS_BACKEND!r}")
    try:
        from sklearn.cluster import KMeans
    except ImportError as e:  # pragma: no cover
        raise ImportError("the 'sklearn' KMeans backend needs scikit-learn") from e
    # the reference's exact call (preprocess_hypergraph.py:150-151, 299-300, 391-392)
    lab = KMeans(n_clusters=n_clusters, random_state=42, n_init=10).fit_predict(x.detach().cpu().numpy())
    return torch.from_numpy(np.asarray(lab)).to(device=x.device, dtype=torch.int64)


# ---------------------------------------------------------------------------------------------------------------------
# HDF5 inputs
# ---------------------------------------------------------------------------------------------------------------------
def load_wsi_data(h5_path: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """wsi/features [N, D] and wsi/positions [N, 2|3] as f32 CPU tensors (:47-62); positions default to zeros [N, 2]."""
    with h5io.open_file(h5_path, "r") as f:
        if "wsi" in f and "features" in f["wsi"]:
            wsi_features = torch.from_numpy(np.asarray(f["wsi"]["features"][:])).float()
        else:
            raise ValueError(f"WSI features not found in {h5_path}")
        if "wsi" in f and "positions" in f["wsi"]:
            wsi_positions = torch.from_numpy(np.asarray(f["wsi"]["positions"][:])).float()
        else:
            wsi_positions = torch.zeros(wsi_features.shape[0], 2, dtype=torch.float32)
            print("WSI positions not found, using dummy positions")
    return wsi_features, wsi_positions


def load_tma_data(h5_path: str) -> Optional[torch.Tensor]:
    """tma/features [N_tma, D] as an f32 CPU tensor, or None (:79-84)."""
    with h5io.open_file(h5_path, "r") as f:
        if "tma" in f and "features" in f["tma"]:
            return torch.from_numpy(np.asarray(f["tma"]["features"][:])).float()
        return None


# ---------------------------------------------------------------------------------------------------------------------
# arithmetic
# ---------------------------------------------------------------------------------------------------------------------
def aggregate_wsi_super_patches(wsi_features: torch.Tensor, wsi_positions: torch.Tensor, num_super_patches: int,
                                lambda_h: float = 1.0, lambda_g: float = 1.0, device: Optional[torch.device] = None,
                                wsi_similarity_matrix: Optional[torch.Tensor] = None):
    """Cluster patches (KMeans on the features, :150-151), mean-pool each cluster (:157-170) and report
    intra-cluster / whole-matrix similarity statistics (:172-197).  Returns
    (super_features, super_positions, stats, K_wsi) on the reference's result device."""
    out_dev = result_device_like_preprocess(wsi_features, device)
    dev = out_dev if out_dev.type == "cuda" else compute_device(wsi_features, wsi_positions)
    F = to_gpu(wsi_features, dev)
    P = to_gpu(wsi_positions, dev)
    N = F.shape[0]
    K = to_gpu(wsi_similarity_matrix, dev) if wsi_similarity_matrix is not None else \
        ops.sim_dense_combined(F, P, float(lambda_h), float(lambda_g))
    labels = _kmeans_labels(F, num_super_patches)
    seg = ops.segment_sort(labels, num_super_patches)
    counts = seg.counts.cpu()
    if int(counts.min()) == 0:
        raise ValueError(f"Cluster {int(torch.argmin(counts))} is empty")
    super_f = ops.segment_mean(F, seg)
    super_p = ops.segment_mean(P, seg)
    # mean off-diagonal similarity inside every cluster with more than one member (:175-184); each cluster's mean is an
    # f32 `.item()` upstream, their average a float64 np.mean
    intra = ops.segment_offdiag_mean(K, seg).to(torch.float32).cpu().numpy()
    intra = intra[~np.isnan(intra)]
    stats = {"num_original_patches": int(N), "num_super_patches": int(num_super_patches),
             "avg_intra_cluster_similarity": float(np.mean(intra.astype(np.float64))) if intra.size else 0.0,
             "wsi_similarity_matrix_stats": ops.array_stats(K)}
    return super_f.to(out_dev), super_p.to(out_dev), stats, K.to(out_dev)


def compute_wsi_tma_similarity(wsi_features: torch.Tensor, wsi_positions: torch.Tensor, tma_features: torch.Tensor,
                               lambda_h: float = 1.0, lambda_g: float = 1.0,
                               device: Optional[torch.device] = None) -> Tuple[torch.Tensor, Dict]:
    """[N_wsi, N_tma] exp(-lambda_h * sum_k (a_k - b_k)^2) + its statistics (:248-265).
    `wsi_positions` and `lambda_g` are accepted and ignored, as in the reference (Appendix A7)."""
    out_dev = result_device_like_preprocess(wsi_features, device)
    dev = out_dev if out_dev.type == "cuda" else compute_device(wsi_features, tma_features)
    S, stats = ops.sim_dense_stats(to_gpu(wsi_features, dev), to_gpu(tma_features, dev), metric="rbf_direct", lam=float(lambda_h))
    return S.to(out_dev), stats


def group_by_similarity(similarity_matrix: torch.Tensor, num_groups: int, method: str = "kmeans"):
    """KMeans over the rows of the similarity matrix (:297-306).  method='knn' is the reference's
    broken branch (Appendix A2) and is not provided."""
    if method != "kmeans":
        raise ValueError(f"Unknown grouping method: {method}")
    dev = compute_device(similarity_matrix)
    labels = _kmeans_labels(to_gpu(similarity_matrix, dev), num_groups)
    sizes = torch.bincount(labels, minlength=num_groups).cpu().tolist()
    stats = {"method": "kmeans", "num_groups": num_groups, "group_sizes": [int(v) for v in sizes]}
    return labels.cpu().numpy().astype(np.int32), stats      # scikit-learn's label dtype: what the reference writes to group_labels


def build_hypergraph_knn_kmeans(wsi_features: torch.Tensor, tma_features: torch.Tensor, group_labels: np.ndarray,
                                k: int = 5, num_hyperedges: int = 10,
                                device: Optional[torch.device] = None) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """k-NN edges + KMeans cliques, undirected dedup, max(0, cosine) weights (:373-433).
    `group_labels` is accepted and ignored, as in the reference (Appendix A7)."""
    out_dev = result_device_like_preprocess(wsi_features, device)
    dev = out_dev if out_dev.type == "cuda" else compute_device(wsi_features, tma_features)
    all_f = torch.cat([to_gpu(wsi_features, dev), to_gpu(tma_features, dev)], dim=0)
    n_total, n_wsi = all_f.shape[0], wsi_features.shape[0]
    if k + 1 > n_total:   # what sklearn's kneighbors raises at :382
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k + 1}, "
                         f"n_samples_fit = {n_total}, n_samples = {n_total}")
    # (a8) Euclidean k-NN, self dropped by identity
    nbr, _ = ops.simtopk(all_f, metric="neg_sq_l2", k=k, exclude_self=True)
    # (a10) cliques of the KMeans hyperedges: every pair inside a cluster, once
    labels = _kmeans_labels(all_f, num_hyperedges)
    seg = ops.segment_sort(labels, num_hyperedges)
    c_lo, c_hi = ops.clique_pairs(seg)
    # (a9) undirected dedup (:403-404): a k-NN pair survives unless the other row emits it too or a clique holds it
    k_lo, k_hi = ops.knn_pairs(nbr, labels)
    code = torch.sort(torch.cat([c_lo, k_lo]) * n_total + torch.cat([c_hi, k_hi])).values      # documented order (A6)
    edge_index = torch.stack([code // n_total, code % n_total], dim=0).contiguous()
    if edge_index.shape[1] == 0:
        edge_index = torch.empty((2, 0), dtype=torch.long, device=dev)
        edge_weights = torch.empty((0,), dtype=torch.float32, device=dev)
    else:
        edge_weights = ops.edge_cosine(all_f, edge_index)
    stats = {"num_nodes": int(n_total), "num_wsi_super_patches": int(n_wsi),
             "num_tma_patches": int(tma_features.shape[0]), "num_edges": int(edge_index.shape[1]),
             "num_hyperedges": int(num_hyperedges), "k": int(k)}
    return edge_index.to(out_dev), edge_weights.to(out_dev), stats


# ---------------------------------------------------------------------------------------------------------------------
# HDF5 output + the two pipelines
# ---------------------------------------------------------------------------------------------------------------------
def save_hypergraph_to_h5(h5_path: str, wsi_super_features: torch.Tensor, wsi_super_positions: torch.Tensor,
                          tma_features: torch.Tensor, edge_index: torch.Tensor, edge_weights: torch.Tensor,
                          group_labels: np.ndarray, stats: Dict, wsi_similarity_matrix: Optional[torch.Tensor] = None,
                          wsi_tma_similarity_matrix: Optional[torch.Tensor] = None):
    """Write the hypergraph/ group (:474-511; layout in h5io).  Existing datasets are replaced (Appendix A4); the stats
    dictionary is made JSON-serialisable first (A3)."""
    def arr(t, dtype=None):
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        return np.ascontiguousarray(a if dtype is None else a.astype(dtype, copy=False))
    with h5io.open_file(h5_path, "a") as f:
        hg = h5io.child(f, "hypergraph")
        ws = h5io.child(hg, "wsi_super")
        h5io.put(ws, "features", arr(wsi_super_features))
        h5io.put(ws, "positions", arr(wsi_super_positions))
        h5io.put(h5io.child(hg, "tma"), "features", arr(tma_features))
        h5io.put(hg, "edge_index", arr(edge_index, np.int64))
        h5io.put(hg, "edge_weights", arr(edge_weights, np.float32))
        h5io.put(hg, "group_labels", arr(group_labels))
        if wsi_similarity_matrix is not None:
            sim = h5io.child(hg, "similarity")
            h5io.put(sim, "wsi_internal", arr(wsi_similarity_matrix))
            sim.attrs["wsi_shape"] = [int(v) for v in wsi_similarity_matrix.shape]
        if wsi_tma_similarity_matrix is not None:
            sim = h5io.child(hg, "similarity")
            h5io.put(sim, "wsi_tma", arr(wsi_tma_similarity_matrix))
            sim.attrs["wsi_tma_shape"] = [int(v) for v in wsi_tma_similarity_matrix.shape]
        hg.attrs["stats"] = h5io.dumps_stats(stats)


def _default_device(device):
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")    # :551-552
    return device


def process_single_file(h5_path: str, num_wsi_super_patches: int = 100, num_groups: int = 10, hypergraph_k: int = 5,
                        num_hyperedges: int = 10, lambda_h: float = 1.0, lambda_g: float = 1.0,
                        device: Optional[torch.device] = None) -> Dict:
    """load -> aggregate -> WSI x TMA similarity -> group -> k-NN + KMeans hypergraph -> save (:554-603)."""
    device = _default_device(device)
    print(f"Processing: {h5_path}")
    wsi_features, wsi_positions = load_wsi_data(h5_path)
    tma_features = load_tma_data(h5_path)
    if tma_features is None:
        print("TMA features not found, skipping hypergraph construction")
        return {"status": "skipped", "reason": "no_tma"}
    wsi_super_features, wsi_super_positions, wsi_stats, wsi_sim_matrix = aggregate_wsi_super_patches(
        wsi_features, wsi_positions, num_wsi_super_patches, lambda_h, lambda_g, device)
    similarity_matrix, sim_stats = compute_wsi_tma_similarity(
        wsi_super_features, wsi_super_positions, tma_features, lambda_h, lambda_g, device)
    group_labels, group_stats = group_by_similarity(similarity_matrix, num_groups, method="kmeans")
    edge_index, edge_weights, hg_stats = build_hypergraph_knn_kmeans(
        wsi_super_features, tma_features, group_labels, hypergraph_k, num_hyperedges, device)
    all_stats = {"wsi_aggregation": wsi_stats, "similarity": sim_stats, "grouping": group_stats, "hypergraph": hg_stats}
    save_hypergraph_to_h5(h5_path, wsi_super_features, wsi_super_positions, tma_features, edge_index, edge_weights,
                          group_labels, all_stats, wsi_similarity_matrix=wsi_sim_matrix,
                          wsi_tma_similarity_matrix=similarity_matrix)
    return all_stats


def _for_each_file(csv_path: str, data_root_dir: str, output_stats_path: Optional[str], desc: str, fn):
    """The shell shared by process_dataset (:644-678) and batch_rebuild_hypergraph (:956-990): CSV with an
    `h5_file_path` column, missing files skipped, a failing file reported and skipped, stats collected."""
    import pandas as pd
    try:
        from tqdm import tqdm
    except ImportError:  # pragma: no cover
        def tqdm(it, **_kw):
            return it
    df = pd.read_csv(csv_path)
    if "h5_file_path" not in df.columns:
        raise ValueError("CSV must contain 'h5_file_path' column")
    all_stats = []
    for idx, row in tqdm(df.iterrows(), total=len(df), desc=desc):
        h5_rel_path = row["h5_file_path"]
        h5_path = os.path.join(data_root_dir, h5_rel_path)
        if not h5io.path_exists(h5_path):
            print(f"File not found: {h5_path}")
            continue
        try:
            stats = fn(h5_path)
            stats["case_id"] = h5io.jsonable(row.get("case_id", f"case_{idx}"))
            stats["h5_path"] = h5_rel_path
            all_stats.append(stats)
        except Exception as e:  # noqa: BLE001 — the reference's per-file error boundary (:667-670)
            print(f"Error processing {h5_path}: {e}")
            import traceback
            traceback.print_exc()
    if output_stats_path:
        with open(output_stats_path, "w") as fh:
            json.dump(h5io.jsonable(all_stats), fh, indent=2)
        print(f"Statistics saved to: {output_stats_path}")
    return all_stats


def process_dataset(csv_path: str, data_root_dir: str, num_wsi_super_patches: int = 100, num_groups: int = 10,
                    hypergraph_k: int = 5, num_hyperedges: int = 10, lambda_h: float = 1.0, lambda_g: float = 1.0,
                    output_stats_path: Optional[str] = None, device: Optional[torch.device] = None):
    """process_single_file over every row of the CSV (:644-678)."""
    return _for_each_file(csv_path, data_root_dir, output_stats_path, "Processing files",
                          lambda p: process_single_file(p, num_wsi_super_patches, num_groups, hypergraph_k, num_hyperedges,
                                                        lambda_h, lambda_g, device))


def load_similarity_matrices(h5_path: str) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(hypergraph/similarity/wsi_internal, hypergraph/similarity/wsi_tma) as f32 CPU tensors, None when absent (:741-755)."""
    with h5io.open_file(h5_path, "r") as f:
        wsi_sim_matrix = None
        wsi_tma_sim_matrix = None
        if "hypergraph" in f and "similarity" in f["hypergraph"]:
            sim_group = f["hypergraph"]["similarity"]
            if "wsi_internal" in sim_group:
                wsi_sim_matrix = torch.from_numpy(np.asarray(sim_group["wsi_internal"][:])).float()
            if "wsi_tma" in sim_group:
                wsi_tma_sim_matrix = torch.from_numpy(np.asarray(sim_group["wsi_tma"][:])).float()
        return wsi_sim_matrix, wsi_tma_sim_matrix


def rebuild_hypergraph_from_similarity(h5_path: str, num_wsi_super_patches: int = None, num_groups: int = None,
                                       hypergraph_k: int = 5, num_hyperedges: int = 10,
                                       threshold_median_ratio: float = None,
                                       device: Optional[torch.device] = None) -> Dict:
    """Rebuild the hypergraph from the stored similarity matrices with other parameters (:795-916)."""
    device = _default_device(device)
    print(f"Rebuilding hypergraph from stored similarity matrices: {h5_path}")
    wsi_features, wsi_positions = load_wsi_data(h5_path)
    tma_features = load_tma_data(h5_path)
    if tma_features is None:
        raise ValueError("TMA features not found")
    wsi_sim_matrix, wsi_tma_sim_matrix = load_similarity_matrices(h5_path)
    if wsi_sim_matrix is None:
        print("WSI similarity matrix not found, recomputing...")
        wsi_sim_matrix = compute_combined_similarity(wsi_features, wsi_positions, lambda_h=1.0, lambda_g=1.0)

    wsi_stats = {}
    if num_wsi_super_patches is not None:
        wsi_super_features, wsi_super_positions, wsi_stats, _ = aggregate_wsi_super_patches(
            wsi_features, wsi_positions, num_wsi_super_patches, lambda_h=1.0, lambda_g=1.0, device=device,
            wsi_similarity_matrix=wsi_sim_matrix)
        similarity_matrix, sim_stats = compute_wsi_tma_similarity(
            wsi_super_features, wsi_super_positions, tma_features, lambda_h=1.0, lambda_g=1.0, device=device)
    else:
        with h5io.open_file(h5_path, "r") as f:
            if "hypergraph" in f and "wsi_super" in f["hypergraph"]:
                wsi_super_features = torch.from_numpy(np.asarray(f["hypergraph"]["wsi_super"]["features"][:])).float().to(device)
                wsi_super_positions = torch.from_numpy(np.asarray(f["hypergraph"]["wsi_super"]["positions"][:])).float().to(device)
            else:
                raise ValueError("WSI super patches not found and num_wsi_super_patches not specified")
        if wsi_tma_sim_matrix is not None and wsi_tma_sim_matrix.shape[0] == wsi_super_features.shape[0]:
            similarity_matrix = wsi_tma_sim_matrix.to(device)
            sim_stats = ops.array_stats(to_gpu(similarity_matrix, compute_device(similarity_matrix)))      # :849-855
        else:
            similarity_matrix, sim_stats = compute_wsi_tma_similarity(
                wsi_super_features, wsi_super_positions, tma_features, lambda_h=1.0, lambda_g=1.0, device=device)

    if num_groups is not None:
        group_labels, group_stats = group_by_similarity(similarity_matrix, num_groups, method="kmeans")
    else:
        with h5io.open_file(h5_path, "r") as f:
            if "hypergraph" in f and "group_labels" in f["hypergraph"]:
                group_labels = np.asarray(f["hypergraph"]["group_labels"][:])
                group_stats = {"method": "existing", "num_groups": int(len(np.unique(group_labels)))}
            else:
                raise ValueError("Group labels not found and num_groups not specified")

    edge_index, edge_weights, hg_stats = build_hypergraph_knn_kmeans(
        wsi_super_features, tma_features, group_labels, hypergraph_k, num_hyperedges, device)

    if threshold_median_ratio is not None:
        # edge-weight median filter (:885-897): torch.median = lower median, here a device radix select
        w_dev = to_gpu(edge_weights, compute_device(edge_weights))
        median_weight = ops.lower_median(w_dev).item()
        threshold = median_weight * threshold_median_ratio
        mask = edge_weights >= threshold
        edge_index = edge_index[:, mask]
        edge_weights = edge_weights[mask]
        hg_stats["num_edges_after_threshold"] = int(edge_weights.shape[0])
        hg_stats["threshold"] = threshold
        hg_stats["threshold_ratio"] = threshold_median_ratio

    all_stats = {"wsi_aggregation": wsi_stats if num_wsi_super_patches is not None else {}, "similarity": sim_stats,
                 "grouping": group_stats, "hypergraph": hg_stats}
    save_hypergraph_to_h5(h5_path, wsi_super_features, wsi_super_positions, tma_features, edge_index, edge_weights,
                          group_labels, all_stats, wsi_similarity_matrix=wsi_sim_matrix,
                          wsi_tma_similarity_matrix=similarity_matrix)
    return all_stats


def batch_rebuild_hypergraph(csv_path: str, data_root_dir: str, num_wsi_super_patches: int = None, num_groups: int = None,
                             hypergraph_k: int = 5, num_hyperedges: int = 10, threshold_median_ratio: float = None,
                             output_stats_path: Optional[str] = None, device: Optional[torch.device] = None):
    """rebuild_hypergraph_from_similarity over every row of the CSV (:956-990)."""
    return _for_each_file(csv_path, data_root_dir, output_stats_path, "Rebuilding hypergraphs",
                          lambda p: rebuild_hypergraph_from_similarity(p, num_wsi_super_patches, num_groups, hypergraph_k,
                                                                       num_hyperedges, threshold_median_ratio, device))


def _main(argv=None):
    """`python -m ... preprocess_hypergraph` with the reference's flags (:681-723)."""
    import argparse
    ap = argparse.ArgumentParser(description="Preprocess hypergraph data")
    ap.add_argument("--csv_path", type=str, required=True)
    ap.add_argument("--data_root_dir", type=str, required=True)
    ap.add_argument("--num_wsi_super_patches", type=int, default=100)
    ap.add_argument("--num_groups", type=int, default=10)
    ap.add_argument("--hypergraph_k", type=int, default=5)
    ap.add_argument("--num_hyperedges", type=int, default=10)
    ap.add_argument("--lambda_h", type=float, default=1.0)
    ap.add_argument("--lambda_g", type=float, default=1.0)
    ap.add_argument("--output_stats", type=str, default=None)
    ap.add_argument("--device", type=str, default="auto")
    ap.add_argument("--kmeans_backend", type=str, default=None, choices=["device", "sklearn"],
                    help="not a reference flag: where KMeans runs (default: MMF_KMEANS_BACKEND or 'device'; same labels)")
    a = ap.parse_args(argv)
    if a.kmeans_backend:
        set_kmeans_backend(a.kmeans_backend)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu") if a.device == "auto" else torch.device(a.device)
    process_dataset(a.csv_path, a.data_root_dir, a.num_wsi_super_patches, a.num_groups, a.hypergraph_k, a.num_hyperedges,
                    a.lambda_h, a.lambda_g, a.output_stats, device)


if __name__ == "__main__":
    _main()
