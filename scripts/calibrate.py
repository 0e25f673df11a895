"""What this box reaches on vendor code: the practical ceilings the roofline fractions can be read against.
  * hipBLASLt f16 / bf16 GEMM 8192^3 (torch.matmul), random and all-zero operands (the clock the chip holds depends on
    how many bits toggle) — the matrix-core ceiling of a kernel that does nothing but MFMA;
  * f32 GEMM 8192^3 — the f32 matrix pipe;
  * HBM: device copy (read + write), fill (write only), sum (read only) of 4 GiB.
Prints one JSON object; profiles/rNN_calibration.json is a copy."""
import json, sys, time, torch
dev = torch.device('cuda')


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
n = 8192
for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
    a, b = torch.randn((n, n), device=dev).to(dt), torch.randn((n, n), device=dev).to(dt)
    t = timed(lambda: torch.matmul(a, b.t()))
    out[f"gemm {name} 8192^3 random"] = {"ms": t * 1e3, "TFLOP/s": 2 * n ** 3 / t / 1e12, "frac of 2500": 2 * n ** 3 / t / 2.5e15}
    a.zero_(); b.zero_()
    t = timed(lambda: torch.matmul(a, b.t()))
    out[f"gemm {name} 8192^3 zeros"] = {"ms": t * 1e3, "TFLOP/s": 2 * n ** 3 / t / 1e12, "frac of 2500": 2 * n ** 3 / t / 2.5e15}
# the scan's own shape: 262144 x 512 against a 16384-row slice of itself, output f16 (the GEMM has to WRITE it)
x = torch.randn((262144, 512), device=dev).half()
y = x[:16384]
t = timed(lambda: torch.matmul(x, y.t()), reps=5)
out["gemm f16 262144 x 16384 x 512 (scan shape, output written)"] = {"ms": t * 1e3, "TFLOP/s": 2 * 262144 * 16384 * 512 / t / 1e12,
                                                                       "frac of 2500": 2 * 262144 * 16384 * 512 / t / 2.5e15}
del x, y
a, b = torch.randn((n, n), device=dev), torch.randn((n, n), device=dev)
torch.backends.cuda.matmul.allow_tf32 = False
t = timed(lambda: torch.matmul(a, b.t()), reps=5)
out["gemm f32 8192^3 random"] = {"ms": t * 1e3, "TFLOP/s": 2 * n ** 3 / t / 1e12, "frac of 157.3": 2 * n ** 3 / t / 157.3e12}
del a, b
nb = 4 << 30
src = torch.empty(nb // 4, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
t = timed(lambda: dst.copy_(src))
out["hbm copy 4 GiB (read + write)"] = {"ms": t * 1e3, "GB/s": 2 * nb / t / 1e9, "frac of 8000": 2 * nb / t / 8e12}
t = timed(lambda: dst.fill_(1.0))
out["hbm fill 4 GiB (write only)"] = {"ms": t * 1e3, "GB/s": nb / t / 1e9, "frac of 8000": nb / t / 8e12}
t = timed(lambda: src.sum())
out["hbm sum 4 GiB (read only)"] = {"ms": t * 1e3, "GB/s": nb / t / 1e9, "frac of 8000": nb / t / 8e12}
print(json.dumps(out, indent=1))
