import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device('cuda')
Y = make_rows(0, 262144, 512, dev)
for rows, cols in ((32768, 262144), (65536, 65536), (65536, 262144), (131072, 262144)):
    X, Yc = Y[:rows], Y[:cols]
    for splits in (1, 2, 4, 8):
        best = 1e9
        for it in range(3):
            i, v, st = mmf.simtopk(X, Yc, metric='cosine', k=5, exclude_self=True, col_splits=splits, return_stats=True, profile=True)
            best = min(best, st['scan_ms'])
        tf = 2.0 * rows * cols * 512 / (best * 1e-3) / 1e12
        print("rows=%d cols=%d splits=%d grid=%d scan=%.2f ms %.0f TF cand/row=%.1f fb=%d" % (rows, cols, st['col_splits'], st['scan_grid'], best, tf, st['candidates'] / rows, st['fallback_rows']), flush=True)
