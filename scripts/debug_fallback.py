import sys, torch, numpy as np
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
dev = torch.device('cuda')
for n, d in [(4096, 128), (65536, 512)]:
    X = make_rows(0, n, d, dev)
    for splits in (1, 4):
        for k, prec in ((5, 'fast'), (7, 'fast'), (5, 'fast_bf16')):
            i, v, st = mmf.simtopk(X, metric='cosine', k=k, precision=prec, col_splits=splits, return_stats=True, profile=True)
            print(prec, "n=%d d=%d splits=%d k=%d: fallback=%d ovf=%d short=%d cand/row=%.2f scan_ms=%.2f" % (
                n, d, st['col_splits'], k, st['fallback_rows'], st['overflow_rows'], st['short_rows'], st['candidates'] / n, st['scan_ms']), flush=True)
