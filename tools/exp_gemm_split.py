"""Dev tool: the aligned dense-layer products (gemm2_kernel) with exact-fp32 matrix instructions or with the bf16 triple split
(ops.PREC_BF16X3, the training default; `exact` on the command line selects ops.PREC_F32, the exact-fp32 instruction): ms per product at
the inner-light net's backward size, and the error of both against fp64 on operands with a gradient-like scale.
    python tools/exp_gemm_split.py [exact]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensoflow_amd import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    exact = len(sys.argv) > 1 and sys.argv[1] == "exact"
    tag = "exact fp32" if exact else "bf16x3 split"
    ops.LINEAR_PRECISION = ops.PREC_F32 if exact else ops.PREC_BF16X3
    # accuracy: 8 192 rows, activations O(1), gradients at 1e-9 (a 1e-4 loss weight over 1e5 samples) with a heavy tail
    n, K, N = 8192, 256, 256
    x = torch.randn(n, K, generator=g)
    w = torch.randn(N, K, generator=g) / 16
    gy = torch.randn(n, N, generator=g) * 1e-9 * torch.exp(3 * torch.randn(n, 1, generator=g))
    y = ops.linear_fwd(x.to(dev), w.to(dev), None)
    ref = x.double() @ w.double().T
    print(f"{tag}: forward   max |err| / max |ref| = {float((y.cpu().double() - ref).abs().max() / ref.abs().max()):.2e}")
    gx, gw, _ = ops.linear_bwd(x.to(dev), w.to(dev), y, gy.to(dev), need_gb=False)
    rgx, rgw = gy.double() @ w.double(), gy.double().T @ x.double()
    print(f"{tag}: data grad max |err| / max |ref| = {float((gx.cpu().double() - rgx).abs().max() / rgx.abs().max()):.2e}")
    print(f"{tag}: weight grad max |err| / max |ref| = {float((gw.cpu().double() - rgw).abs().max() / rgw.abs().max()):.2e}")
    # speed: the inner-light net's hidden layers over a material training step's hit rays
    n = 236_000
    x = torch.randn(n, K, generator=g).to(dev)
    wd = w.to(dev)
    gyd = torch.randn(n, N, generator=g).to(dev)
    yd = ops.linear_fwd(x, wd, None)
    fl = 2.0 * n * K * N
    t = timed(lambda: ops.linear_fwd(x, wd, None))
    print(f"{tag}: forward {t:.3f} ms = {fl / t / 1e9:.1f} TF/s")
    t = timed(lambda: ops.linear_bwd(x, wd, yd, gyd, need_gw=False, need_gb=False))
    print(f"{tag}: data gradient (+ activation pass) {t:.3f} ms")
    t = timed(lambda: ops.linear_bwd(x, wd, yd, gyd, need_gx=False, need_gb=False))
    print(f"{tag}: weight gradient (+ activation pass) {t:.3f} ms")


if __name__ == "__main__":
    main()
