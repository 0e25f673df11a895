"""One KMeans fit under rocprofv3 --kernel-trace: which kernels the time goes to."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from importlib import import_module
km = import_module("multimodal_fusion_amd.kmeans")
rng = np.random.RandomState(0)
N, D, S = 16384, 512, 100
cent = rng.randn(200, D).astype(np.float32)
W = torch.from_numpy((cent[rng.randint(0, 200, N)] * 0.3 + 0.05 * rng.randn(N, D)).astype(np.float32)).cuda()
km.kmeans_fit_predict(W[:2048], 8, n_init=1)
torch.cuda.synchronize()
km.kmeans_fit_predict(W, S)
torch.cuda.synchronize()
