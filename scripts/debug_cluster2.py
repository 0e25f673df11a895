import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import multimodal_fusion_amd as mmf
from conftest import unit_rows
def rnd(n, d, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(n, d) * scale).astype(np.float32)
X = unit_rows(3000, 128, 77).numpy()
X[100:160] = X[100] + 1e-4 * rnd(60, 128, 78)
i, v, st = mmf.simtopk(torch.tensor(X).cuda(), metric='cosine', k=5, precision='fast', col_splits=1, return_stats=True)
torch.cuda.synchronize()
print(st)
S = X.astype(np.float64); S /= np.linalg.norm(S, axis=1, keepdims=True); C = S @ S.T
for r in (5, 100, 300, 2000):
    srt = np.sort(C[r])[::-1]
    print(r, "top8", srt[:8], "t6*65536", srt[5] * 65536, "c_100", C[r, 100], "count within 0.001 of t6:", (C[r] >= srt[5] - 0.001).sum())
