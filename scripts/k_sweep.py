import sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
X = make_rows(0, 65536, 512, torch.device('cuda'))
for k in (5, 11, 15, 16, 19, 20):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i, v, st = mmf.simtopk(X, metric='cosine', k=k, return_stats=True, profile=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("k=%d: wall %.2f ms scan %.2f rerank %.2f fb %d precision_used %d cand/row %.1f" % (k, dt, st['scan_ms'], st['rerank_ms'], st['fallback_rows'], st['precision_used'], st['candidates'] / 65536))
