"""Dev: the dense-layer kernels at TF_PREC_F16X3 (the older register-staged kernel's operand-split path) against the exact-fp32 ones."""
import sys, time, os
import torch
sys.path.insert(0, ".")
from tensoflow_amd import ops

def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 350000
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
for K, N in ((256, 256), (128, 256), (112, 256)):
    x = torch.randn(rows, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g); gy = torch.randn(rows, N, device=dev, generator=g)
    fl = 2.0 * rows * K * N
    for prec, name in ((ops.PREC_F32, "f32  "), (ops.PREC_F16X3, "f16x3")):
        y = ops.linear_fwd(x, w, b, ops.ACT_RELU, 0.0, precision=prec)
        t = bench(lambda: ops.linear_fwd(x, w, b, ops.ACT_RELU, 0.0, precision=prec))
        t3 = bench(lambda: ops.linear_bwd(x, w, y, gy, ops.ACT_RELU, 0.0, precision=prec))
        ref = torch.relu(x[:4096].double() @ w.double().t() + b.double())
        print(f"{name} rows {rows} K {K} N {N}: fwd {t*1e3:.3f} ms = {fl/t/1e12:.1f} TF/s | full bwd {t3*1e3:.3f} ms = {2*fl/t3/1e12:.1f} TF/s | fwd err {float((y[:4096].double()-ref).abs().max()):.2e}")
