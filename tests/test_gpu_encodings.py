"""tf_ide5_fwd / tf_ide5_bwd / tf_posenc_fwd (round 4) against the torch compositions they replace: the integrated directional
encoding of utils/ref_utils.py:53-117 evaluated in fp64 Horner form on the reference's fp32 coefficient table
(encodings._ide5_wide) and its autograd gradient wrt the direction and kappa_inv; the positional encoding of
utils/network_utils.py:38-50.  (Both compositions are pinned to the reference by tests/golden/encodings.npz on CPU.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected but no GPU is visible")
    return torch.device("cuda:0")


def test_ide5_broadcast_kappa_and_argument_checks(dev):
    """A kappa_inv that broadcasts against the rows ([1,1], 0-dim, a Python 0) is valid in the torch composition the kernel replaced;
    the fused path expands it to one value per row instead of reading past its end, and Ide5Fn rejects what it cannot take."""
    from tensoflow_amd import encodings as E
    g = torch.Generator().manual_seed(0)
    xyz = torch.nn.functional.normalize(torch.randn(257, 3, generator=g), dim=-1).to(dev)
    full = E.ide5(xyz, torch.full((257, 1), 0.3, device=dev))
    k11 = torch.full((1, 1), 0.3, device=dev, requires_grad=True)
    out = E.ide5(xyz, k11)
    assert torch.equal(out, full)
    gk, = torch.autograd.grad(out.sum(), [k11])
    kf = torch.full((257, 1), 0.3, device=dev, requires_grad=True)
    gf, = torch.autograd.grad(E.ide5(xyz, kf).sum(), [kf])
    assert gk.shape == (1, 1) and abs(float(gk) - float(gf.sum())) < 1e-4 * abs(float(gf.sum()))
    assert torch.equal(E.ide5(xyz, torch.tensor(0.3, device=dev)), full)
    assert torch.equal(E.ide5(xyz, 0), E.ide5(xyz, torch.zeros(257, 1, device=dev)))
    with pytest.raises(RuntimeError, match="one fp32 value per row"):
        E.Ide5Fn.apply(xyz, torch.zeros(5, 1, device=dev))
    with pytest.raises(RuntimeError, match=r"fp32 \[n,3\]"):
        E.Ide5Fn.apply(xyz[:, :2], None)
    # kappa on another device / dtype: the torch composition serves it (no kernel reads it)
    half = E.ide5(xyz, torch.full((257, 1), 0.3, device=dev, dtype=torch.float64))
    assert float((half.float() - full).abs().max()) < 1e-5


@pytest.mark.parametrize("n,with_kappa", [(5000, True), (777, False), (1, True)])
def test_ide5_kernel_forward_and_backward(dev, n, with_kappa):
    from tensoflow_amd import encodings as E
    g = torch.Generator().manual_seed(n)
    xyz = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    xyz[: min(n, 3)] = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])[: min(n, 3)]      # poles and axes
    kap = torch.rand(n, 1, generator=g) * 0.9 + 0.05 if with_kappa else torch.zeros(n, 1)
    w = torch.randn(n, 72, generator=g)
    x1 = xyz.to(dev).requires_grad_(True)
    k1 = kap.to(dev).requires_grad_(with_kappa)
    out = E.ide5(x1, k1)                                           # the kernel pair (CUDA, 2-D)
    assert out.shape == (n, 72) and out.dtype == torch.float32
    gx, = torch.autograd.grad((out * w.to(dev)).sum(), [x1], retain_graph=with_kappa)
    x2 = xyz.double().requires_grad_(True)
    k2 = kap.double().requires_grad_(with_kappa)
    # fp64 reference: the same Horner evaluation in torch (CPU, double end to end)
    mat, ms, ls = E._ide_tables()
    mat64 = torch.from_numpy(mat.astype("float64"))
    X, Y, Z = x2[:, 0:1], x2[:, 1:2], x2[:, 2:3]
    poly = mat64[16].expand(n, 36)
    for kk in range(15, -1, -1):
        poly = poly * Z + mat64[kk]
    re, im = [torch.ones_like(X)], [torch.zeros_like(X)]
    for _ in range(16):
        re, im = re + [re[-1] * X - im[-1] * Y], im + [re[-1] * Y + im[-1] * X]
    msl = torch.from_numpy(ms)
    re, im = torch.cat(re, -1)[:, msl], torch.cat(im, -1)[:, msl]
    att = torch.exp(-torch.from_numpy(0.5 * ls * (ls + 1)).double() * k2)
    ref = torch.cat([re * poly * att, im * poly * att], -1)
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
    grads = torch.autograd.grad((ref * w.double()).sum(), [x2] + ([k2] if with_kappa else []))
    scale = float(grads[0].abs().max())
    assert float((gx.cpu().double() - grads[0]).abs().max()) < 2e-6 * scale
    if with_kappa:
        gk, = torch.autograd.grad((out * w.to(dev)).sum(), [k1])
        assert gk.shape == k1.shape
        assert float((gk.cpu().double() - grads[1]).abs().max()) < 2e-6 * float(grads[1].abs().max())


def test_posenc_kernel(dev):
    from tensoflow_amd import encodings as E
    x = (torch.rand(4097, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1)
    for nf in (0, 4, 6, 8):
        got = E.posenc(x.to(dev), nf)                              # kernel (CUDA, no grad)
        ref = E.posenc(x.double(), nf)                             # torch composition, double
        assert got.shape == ref.shape and float((got.cpu().double() - ref).abs().max()) < 2e-5      # sin / cos of arguments up to 128 in fp32
    xg = x.to(dev).requires_grad_(True)
    assert E.posenc(xg, 2).requires_grad                           # differentiable inputs keep the torch composition


@pytest.mark.parametrize("shape", [(1, 36, 32, 32), (1, 36, 300, 1), (1, 4, 7, 13), (1, 1, 1, 5)])
def test_tv_loss_kernels(dev, shape):
    """tf_tv_fwd / tf_tv_bwd against TVLoss.forward (other_field.py:170-191) written out in torch (double) and its autograd gradient."""
    from tensoflow_amd.network.fields import TVLoss
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    xd = x.to(dev).requires_grad_(True)
    got = TVLoss(0.7)(xd)
    (g,) = torch.autograd.grad(got * 3.0, [xd])
    x64 = x.double().requires_grad_(True)
    b, c, h, w = shape
    tot = 0.0
    if c * (h - 1) * w:
        tot = tot + torch.pow(x64[:, :, 1:, :] - x64[:, :, :h - 1, :], 2).sum() / (c * (h - 1) * w)
    if c * h * (w - 1):
        tot = tot + torch.pow(x64[:, :, :, 1:] - x64[:, :, :, :w - 1], 2).sum() / (c * h * (w - 1))
    ref = 0.7 * 2 * tot / b
    (gr,) = torch.autograd.grad(ref * 3.0, [x64])
    assert abs(float(got) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert float((g.cpu().double() - gr).abs().max()) < 1e-5 * max(1e-6, float(gr.abs().max()))
    again = TVLoss(0.7)(xd)
    assert float(again) == float(got)                       # fixed-order reduction: identical bits run to run


def test_tv_loss_of_six_grids_in_one_accumulation(dev):
    """TensoSDF.TV_loss_sdf (fields.py:133-138) through TvLossSumFn (round 5: tf_tv_fwd + tf_tv_finish per grid into ONE device scalar)
    against the per-grid TVLoss sum in double precision: value, the gradient of every grid, and identical bits run to run."""
    from tensoflow_amd.autograd import TvLossSumFn
    from tensoflow_amd.network.fields import TVLoss
    gen = torch.Generator().manual_seed(21)
    shapes = [(1, 36, 48, 48), (1, 36, 48, 1), (1, 36, 40, 48), (1, 36, 40, 1), (1, 36, 48, 40), (1, 36, 48, 1)]
    xs = [torch.randn(*s, generator=gen) for s in shapes]
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    got = TvLossSumFn.apply(0.3, *xd)
    grads = torch.autograd.grad(got * 2.0, xd)
    x64 = [x.double().requires_grad_(True) for x in xs]
    ref = sum(TVLoss(0.3)(x) for x in x64)
    gref = torch.autograd.grad(ref * 2.0, x64)
    assert abs(float(got) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    for g, gr in zip(grads, gref):
        assert float((g.cpu().double() - gr).abs().max()) < 1e-5 * float(gr.abs().max())
    assert float(TvLossSumFn.apply(0.3, *xd)) == float(got)
