"""tf_linear_fwd / tf_linear_bwd (the dense layers of the training direction) against torch.nn.functional on the same device."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_act(z, act, p):
    from tensoflow_amd import ops
    return {ops.ACT_NONE: lambda: z, ops.ACT_RELU: lambda: torch.relu(z), ops.ACT_SOFTPLUS: lambda: F.softplus(z, beta=p),
            ops.ACT_SIGMOID: lambda: torch.sigmoid(z), ops.ACT_EXP_CLAMP: lambda: torch.exp(z.clamp(max=p))}[act]()


@pytest.fixture(params=["f16x3", "f32", "bf16x3"])
def linear_precision(request):
    """The three operand arithmetics of tf_linear_*: the bf16 triple split (the training default: fp32-grade, fp32's range), the exact
    fp32 MFMA and the f16x3 split -- the same 2e-5 bar for all."""
    from tensoflow_amd import ops
    keep = ops.LINEAR_PRECISION
    ops.LINEAR_PRECISION = {"f16x3": ops.PREC_F16X3, "f32": ops.PREC_F32, "bf16x3": ops.PREC_BF16X3}[request.param]
    yield request.param
    ops.LINEAR_PRECISION = keep


# shapes of every kernel family behind tf_linear_*: the register-staged tiles (odd strides: 111, 123, 129 columns), the LDS-DMA pipeline
# (strides that are multiples of four: ragged row counts, a ragged reduction of 108, two column blocks), the streaming kernels of layers
# with <= 4 outputs
@pytest.mark.parametrize("n,K,N", [(1, 3, 1), (777, 111, 256), (4096, 256, 129), (2048, 108, 128), (300, 128, 3), (5000, 123, 256), (130, 256, 256),
                                   (3333, 256, 256), (5000, 128, 128), (1029, 72, 256), (5000, 256, 3), (2048, 108, 1), (999, 128, 4), (1500, 64, 2),
                                   (500, 32, 32), (64, 128, 128), (20000, 36, 128), (77, 4, 3), (1, 1024, 2)])
def test_linear_fwd_bwd_matches_torch(n, K, N, linear_precision):
    from tensoflow_amd import ops
    from tensoflow_amd.autograd import LinearActFn
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(n + K + N)
    for act, p in ((ops.ACT_NONE, 0.0), (ops.ACT_RELU, 0.0), (ops.ACT_SOFTPLUS, 100.0), (ops.ACT_SIGMOID, 0.0), (ops.ACT_EXP_CLAMP, 0.5)):
        x = torch.randn(n, K, generator=g).to(dev).requires_grad_(True)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).requires_grad_(True)
        b = (0.1 * torch.randn(N, generator=g)).to(dev).requires_grad_(True)
        gy = torch.randn(n, N, generator=g).to(dev)
        y = LinearActFn.apply(x, w, b, act, p)
        y.backward(gy)
        got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        x.grad = w.grad = b.grad = None
        yr = _ref_act(F.linear(x.double(), w.double(), b.double()), act, p)
        yr.backward(gy.double())
        ref = (yr.detach(), x.grad, w.grad, b.grad)
        for name, a, r in zip(("y", "gx", "gw", "gb"), got, ref):
            scale = float(r.abs().max()) + 1e-12
            if name == "gb":      # a column sum of n signed terms can cancel to almost nothing (N = 1: there is no other column to set the scale)
                scale = max(scale, 1e-3 * float(gy.abs().sum(0).max()))
            assert float((a.double() - r.double()).abs().max()) / scale < 2e-5, (act, name, n, K, N)


def test_linear_device_side_row_count(linear_precision):
    """n_dev: rows beyond the device-side count are neither computed nor differentiated (compacted hit lists, no host sync)."""
    from tensoflow_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    n, K, N, nv = 1000, 123, 256, 437
    x, w, b = torch.randn(n, K, device=dev), torch.randn(N, K, device=dev) / 11, torch.randn(N, device=dev)
    cnt = torch.tensor([nv], dtype=torch.int64, device=dev)
    y = ops.linear_fwd(x, w, b, ops.ACT_RELU, n_dev=cnt)
    ref = torch.relu(F.linear(x[:nv], w, b))
    assert torch.allclose(y[:nv], ref, atol=1e-4)
    gy = torch.randn(n, N, device=dev)
    yfull = torch.relu(F.linear(x, w, b))
    gx, gw, gb = ops.linear_bwd(x, w, yfull, gy, ops.ACT_RELU, n_dev=cnt)
    gz = gy[:nv] * (yfull[:nv] > 0)
    assert torch.allclose(gw, gz.t() @ x[:nv], atol=2e-3, rtol=1e-4) and torch.allclose(gb, gz.sum(0), atol=1e-3)
    assert torch.allclose(gx[:nv], gz @ w, atol=1e-4)


@pytest.mark.parametrize("K,N", [(256, 256), (256, 3), (128, 128)])
def test_linear_device_side_row_count_aligned_kernels(K, N):
    """The same contract on the LDS-DMA pipeline and on the streaming kernels (exact fp32)."""
    from tensoflow_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    n, nv = 3000, 1437
    x, w, b = torch.randn(n, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
    cnt = torch.tensor([nv], dtype=torch.int64, device=dev)
    y = torch.full((n, N), 7.0, device=dev)
    y = ops.linear_fwd(x, w, b, ops.ACT_RELU, n_dev=cnt)
    assert torch.allclose(y[:nv], torch.relu(F.linear(x[:nv], w, b)), atol=1e-4)
    gy = torch.randn(n, N, device=dev)
    yfull = torch.relu(F.linear(x, w, b))
    gx, gw, gb = ops.linear_bwd(x, w, yfull, gy, ops.ACT_RELU, n_dev=cnt)
    gz = gy[:nv] * (yfull[:nv] > 0)
    assert torch.allclose(gw, gz.t() @ x[:nv], atol=2e-3, rtol=1e-4) and torch.allclose(gb, gz.sum(0), atol=1e-3)
    assert torch.allclose(gx[:nv], gz @ w, atol=1e-4)
    zero = torch.zeros(1, dtype=torch.int64, device=dev)
    _, gw0, gb0 = ops.linear_bwd(x, w, yfull, gy, ops.ACT_RELU, n_dev=zero)
    assert float(gw0.abs().max()) == 0.0 and float(gb0.abs().max()) == 0.0


def test_mlp_apply_walks_weight_normed_sequentials():
    from tensoflow_amd.autograd import mlp_apply
    from tensoflow_amd.network.fields import _predictor3
    import torch.nn as nn
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    seq = _predictor3(108, 3, nn.Sigmoid()).to(dev)
    x = torch.randn(513, 108, device=dev, requires_grad=True)
    y = mlp_apply(seq, x)
    yr = seq(x)
    assert torch.allclose(y, yr, atol=1e-5)
    gy = torch.randn_like(y)
    g1 = torch.autograd.grad(y, [x] + list(seq.parameters()), gy)
    g2 = torch.autograd.grad(yr, [x] + list(seq.parameters()), gy)
    for a, r in zip(g1, g2):
        assert float((a - r).abs().max()) <= 2e-5 * (float(r.abs().max()) + 1e-9) + 1e-7


def test_bf16_split_keeps_non_finite_operands_in_their_rows():
    """the bf16 triple split (TF_PREC_BF16X3) is named by the caller now, and what it does with non-finite operands is pinned:
    x = hi + mid + lo cannot propagate an Inf as the fp32 instruction does (Inf times the other operand's signed residual planes is +Inf
    and -Inf), so every output such an operand reaches is non-finite (NaN or Inf) -- in ITS row only, every other row is untouched and
    agrees with the exact instruction -- while TF_PREC_F32, the exact instruction, returns the IEEE +Inf."""
    from tensoflow_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, K, N = 512, 256, 256
    x = torch.randn(n, K, device=dev)
    w = torch.randn(N, K, device=dev).abs() / 16          # positive weights: an Inf input gives +Inf in every exact-fp32 output of its row
    x[7, 3] = float("inf")
    x[9, 100] = 3.4e38                                       # finite in fp32, rounds up to the bf16 Inf
    x[11, 5] = float("nan")
    yb = ops.linear_fwd(x, w, None, precision=ops.PREC_BF16X3)
    ye = ops.linear_fwd(x, w, None, precision=ops.PREC_F32)
    # (row 9: 3.4e38 is finite in fp32 and so is its exact product row; it rounds up to the bf16 Inf, so the split loses that row too --
    # operands of the split must stay below 3.3895e38 in magnitude, as the header says)
    assert torch.isinf(ye[7]).all() and (ye[7] > 0).all() and torch.isnan(ye[11]).all() and torch.isfinite(ye[9]).all()
    for r in (7, 9, 11):
        assert not torch.isfinite(yb[r]).any(), r
    clean = torch.ones(n, dtype=torch.bool, device=dev)
    clean[[7, 9, 11]] = False
    clean_e = clean.clone()
    assert torch.isfinite(yb[clean]).all() and torch.isfinite(ye[clean_e]).all()
    assert float((yb[clean] - ye[clean]).abs().max()) < 2e-5 * float(ye[clean].abs().max())
