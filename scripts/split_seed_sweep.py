"""Scan time of one rank of 8 (32768 local rows x 262144 columns) as a function of col_splits, i.e. of the
workgroup granularity; later rounds start from the thresholds earlier ones published."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device('cuda')
N, d = 262144, 512
Y = make_rows(0, N, d, dev)
for P in (8, 4, 1):
    X = Y[: N // P]
    for splits in (0, 1, 2, 4, 8, 16, 32):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            i, v, st = mmf.simtopk(X, Y, metric='cosine', k=5, exclude_self=True, row_offset=0, return_stats=True, profile=True, col_splits=splits)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print("P=%d splits=%d(%d) grid=%d: wall=%.2f scan=%.2f rerank=%.2f cand/row=%.1f fb=%d" % (
            P, splits, st['col_splits'], st['scan_grid'], dt, st['scan_ms'], st['rerank_ms'], st['candidates'] / X.shape[0], st['fallback_rows']), flush=True)
