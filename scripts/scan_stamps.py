import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MMF_SCAN_DEBUG"] = "80"
import multimodal_fusion_amd as mmf
from bench import make_rows
for data in ("gaussian", "clustered"):
    X = make_rows(0, 262144, 512, torch.device("cuda", 0), data=data)
    for rep in range(2):
        _, _, st = mmf.simtopk(X, metric="cosine", k=5, precision="fast", profile=True, return_stats=True)
    print(data, st["scan_ms"], flush=True)
