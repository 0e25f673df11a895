"""Cost of redoing a few flagged rows exactly (MMF_DEBUG_FLAG_ROWS) at the benchmark size."""
import os, sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device('cuda')
N, d = 262144, 512
Y = make_rows(0, N, d, dev)
X = Y[: N // 8]
ref = mmf.simtopk(X, Y, metric='cosine', k=5, row_offset=0)
for f in (0, 1, 4, 16, 17, 128):
    os.environ["MMF_DEBUG_FLAG_ROWS"] = str(f)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i, v, st = mmf.simtopk(X, Y, metric='cosine', k=5, row_offset=0, return_stats=True, profile=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("flagged=%d: wall=%.2f ms fallback=%.2f ms rows=%d parity=%s" % (f, dt, st['fallback_ms'], st['fallback_rows'],
          torch.equal(i, ref[0]) and torch.equal(v, ref[1])), flush=True)
