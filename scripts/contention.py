"""How the scan of one rank of 8 reacts to a co-resident 'collective' kernel (scripts/occupy): W workgroups held
for T ms on another stream, launched just before the scan.  Sweeps the scan's workgroup granularity."""
import ctypes, sys, time, torch
sys.path.insert(0, '.')
import multimodal_fusion_amd as mmf
from bench import make_rows
occ = ctypes.CDLL("scripts/occupy/liboccupy.so")
occ.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda')
N, d, P = 262144, 512, 8
Y = make_rows(0, N, d, dev)
X = Y[: N // P]
sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream()
for wgs, ms in ((0, 0.0), (1, 2.0), (8, 2.0), (32, 0.05), (32, 1.0), (32, 2.0), (32, 4.0)):
    for splits in (2, 8):
        best = 1e9
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if wgs:
                occ.occupy_launch(wgs, ms, sink.data_ptr(), side.cuda_stream)
            i, v = mmf.simtopk(X, Y, metric='cosine', k=5, exclude_self=True, row_offset=0, col_splits=splits)
            torch.cuda.current_stream().synchronize(); dt = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            best = min(best, dt)
        print("occupant %2d WGs x %.1f ms | col_splits=%d : step %.2f ms" % (wgs, ms, splits, best), flush=True)
