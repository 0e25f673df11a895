import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import multimodal_fusion_amd as mmf
from conftest import unit_rows
def rnd(n, d, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(n, d) * scale).astype(np.float32)
X = unit_rows(3000, 128, 77).numpy()
for tag, mod in (("plain", False), ("cluster", True)):
    Y = X.copy()
    if mod: Y[100:160] = Y[100] + 1e-4 * rnd(60, 128, 78)
    for splits in (0, 1, 4, 32):
        for prec in ("fast", "fast_bf16"):
            i, v, st = mmf.simtopk(torch.tensor(Y).cuda(), metric='cosine', k=5, precision=prec, col_splits=splits, return_stats=True)
            print(tag, prec, "splits=%d: fallback=%d ovf=%d short=%d cand/row=%.2f" % (st['col_splits'], st['fallback_rows'], st['overflow_rows'], st['short_rows'], st['candidates'] / 3000), flush=True)
