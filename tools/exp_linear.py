"""Dev tool: tf_linear_fwd / tf_linear_bwd on the shapes of a training step (rows x 256 x 256 etc.), TF/s against the exact-fp32
matrix-core peak (157 TF/s).  usage: python tools/exp_linear.py [rows]"""
import sys
import time

import torch

import os

sys.path.insert(0, ".")
from tensoflow_amd import lib as L  # noqa: E402
if os.environ.get("TF_LIB"):          # dev: an alternative build (tools/build_variant.sh)
    L.LIB_PATH = os.path.abspath(os.environ["TF_LIB"])
from tensoflow_amd import ops  # noqa: E402


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 236000
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    for K, N in ((256, 256), (123, 256), (256, 3), (128, 128), (111, 256)):
        x = torch.randn(rows, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        b = torch.randn(N, device=dev, generator=g)
        gy = torch.randn(rows, N, device=dev, generator=g)
        y = ops.linear_fwd(x, w, b, ops.ACT_RELU, 0.0)
        fl = 2.0 * rows * K * N
        t = bench(lambda: ops.linear_fwd(x, w, b, ops.ACT_RELU, 0.0))
        print(f"rows {rows} K {K} N {N}: fwd {t * 1e3:.3f} ms = {fl / t / 1e12:.1f} TF/s", end="")
        t = bench(lambda: ops.linear_bwd(x, w, y, gy, ops.ACT_RELU, 0.0, need_gx=True, need_gw=False, need_gb=True))
        print(f" | act' + data {t * 1e3:.3f} ms", end="")
        t2 = bench(lambda: ops.linear_bwd(x, w, y, gy, ops.ACT_RELU, 0.0, need_gx=False, need_gw=True, need_gb=True))
        print(f" | act' + weight {t2 * 1e3:.3f} ms", end="")
        t3 = bench(lambda: ops.linear_bwd(x, w, y, gy, ops.ACT_RELU, 0.0, need_gx=True, need_gw=True, need_gb=True))
        print(f" | full bwd {t3 * 1e3:.3f} ms = {2 * fl / t3 / 1e12:.1f} TF/s")
        # correctness against torch (fp32 library GEMM) on a slice
        ref = torch.relu(x[:4096] @ w.t() + b)
        print("    fwd max err", float((y[:4096] - ref).abs().max()))


if __name__ == "__main__":
    main()
