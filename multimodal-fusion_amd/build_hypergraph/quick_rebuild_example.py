"""`quick_rebuild_example.py` of the reference (build_hypergraph/quick_rebuild_example.py:12-58; BASELINE config 1 names
it): rebuild every file's hypergraph from its stored similarity matrices with other parameters.

    python -m multimodal_fusion_amd.build_hypergraph.quick_rebuild_example --csv_path ... --data_root_dir ... [flags]
"""
import argparse

from .preprocess_hypergraph import batch_rebuild_hypergraph, set_kmeans_backend


def main(argv=None):
    ap = argparse.ArgumentParser(description="Quickly rebuild hypergraphs with different parameters")
    ap.add_argument("--csv_path", type=str, required=True, help="CSV with an h5_file_path column")
    ap.add_argument("--data_root_dir", type=str, required=True, help="root directory of the h5 files")
    ap.add_argument("--num_wsi_super_patches", type=int, default=None, help="None keeps the stored super patches")
    ap.add_argument("--num_groups", type=int, default=None, help="None keeps the stored group labels")
    ap.add_argument("--hypergraph_k", type=int, default=5)
    ap.add_argument("--num_hyperedges", type=int, default=10)
    ap.add_argument("--threshold_median_ratio", type=float, default=None, help="edge-weight median filter (None: off)")
    ap.add_argument("--output_stats", type=str, default=None)
    ap.add_argument("--kmeans_backend", type=str, default=None, choices=["device", "sklearn"],
                    help="not a reference flag: where KMeans runs (default: MMF_KMEANS_BACKEND or 'device'; same labels)")
    a = ap.parse_args(argv)
    if a.kmeans_backend:
        set_kmeans_backend(a.kmeans_backend)
    return batch_rebuild_hypergraph(a.csv_path, a.data_root_dir, num_wsi_super_patches=a.num_wsi_super_patches,
                                    num_groups=a.num_groups, hypergraph_k=a.hypergraph_k, num_hyperedges=a.num_hyperedges,
                                    threshold_median_ratio=a.threshold_median_ratio, output_stats_path=a.output_stats)


if __name__ == "__main__":
    main()
