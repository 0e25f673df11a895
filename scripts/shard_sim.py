"""What one rank of a P-GPU job does (minus the all-gather): N/P local rows against all N columns."""
import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device('cuda')
N, d = 262144, 512
Y = make_rows(0, N, d, dev)
for P in (1, 2, 4, 8):
    X = Y[: N // P]
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i, v, st = mmf.simtopk(X, Y, metric='cosine', k=5, exclude_self=True, row_offset=0, return_stats=True, profile=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("P=%d rows=%d: wall=%.2f ms scan=%.2f prep=%.2f rerank=%.2f fallback_rows=%d splits=%d grid=%d  -> speedup vs P=1 scan-only basis" % (
        P, N // P, dt, st['scan_ms'], st['prep_ms'], st['rerank_ms'], st['fallback_rows'], st['col_splits'], st['scan_grid']), flush=True)
