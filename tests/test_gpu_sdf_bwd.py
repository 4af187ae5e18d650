"""tf_sdf_alpha_bwd -- the backward of ShapeRenderer.compute_sdf_alpha (shapeRenderer.py:995-1025 over fields.py:227-260, :262-299)
as one entry point -- against autograd through the torch composition of the same function (autograd.sdf_alpha_composed: HIP gather /
scatter + exact-fp32 dense layers + element-wise torch), which in turn is pinned to the reference's gradients by the `march_grad`
golden (tests/test_gpu_renderers.py).  Every upstream gradient on, ragged sizes, fractional / integer / absent mip levels."""
import pytest
import torch

from conftest import AABB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected but no GPU is visible")
    return torch.device("cuda:0")


def _setup(dev, n, R=32, level_mode="frac", seed=0):
    from tensoflow_amd.synth import random_sdf_state
    g = torch.Generator().manual_seed(seed)
    sd = random_sdf_state(seed=1, R=R)
    planes = [sd[f"sdf_plane.{i}"].to(dev).requires_grad_(True) for i in range(3)]
    lines = [(sd[f"sdf_line.{i}"] * (1.0 + 0.3 * torch.randn(sd[f"sdf_line.{i}"].shape, generator=g))).to(dev).requires_grad_(True) for i in range(3)]
    W1, b1 = sd["sdf_mat.0.weight"].to(dev).requires_grad_(True), sd["sdf_mat.0.bias"].to(dev).requires_grad_(True)
    W2, b2 = sd["sdf_mat.2.weight"].to(dev).requires_grad_(True), sd["sdf_mat.2.bias"].to(dev).requires_grad_(True)
    pts = (torch.rand(n, 3, generator=g) * 1.9 - 0.95).to(dev)
    pts[: min(n, 7)] = torch.tensor([0.9999, -0.9999, 0.5])                   # taps that clamp at the box
    level = {"frac": torch.rand(n, generator=g) * 3.2 - 0.6, "int": torch.randint(0, 3, (n,), generator=g).float(), "none": None}[level_mode]
    level = None if level is None else level.to(dev)
    dists = (0.002 + 0.01 * torch.rand(n, generator=g)).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    inv_s = torch.tensor(20.0, device=dev, requires_grad=True)
    ups = [torch.randn(n, generator=g).to(dev), torch.randn(n, 3, generator=g).to(dev), torch.randn(n, 128, generator=g).to(dev) * 0.1,
           torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev) * 1e-4]
    units = [2.0 / (R - 1)] * 3
    return planes, lines, W1, b1, W2, b2, pts, level, dists, dirs, inv_s, ups, units


@pytest.mark.parametrize("n,level_mode,cos_anneal", [(1000, "frac", 0.5), (129, "int", 1.0), (31, "none", 0.0), (4096, "frac", 0.3)])
def test_sdf_alpha_bwd_matches_composed_autograd(dev, n, level_mode, cos_anneal):
    from tensoflow_amd.autograd import SdfAlphaFn, sdf_alpha_composed
    planes, lines, W1, b1, W2, b2, pts, level, dists, dirs, inv_s, ups, units = _setup(dev, n, level_mode=level_mode)
    params = planes + lines + [W1, b1, W2, b2]
    outs = SdfAlphaFn.apply(pts, level, dists, dirs, inv_s, cos_anneal, AABB, units, 3, *params)
    loss = sum((o * u).sum() for o, u in zip(outs, ups))
    got = torch.autograd.grad(loss, params + [inv_s])
    ref_out = sdf_alpha_composed(planes, lines, W1, b1, W2, b2, pts, level, dists, dirs, inv_s, cos_anneal, AABB, units, 3)
    # the composition evaluates alpha / grad / feat / sdf / nh in the same order as SdfAlphaFn returns them
    for o, r, name in zip(outs, ref_out, ("alpha", "grad", "feat", "sdf", "nh")):
        tol = 2e-3 if name == "nh" else 1e-4                                  # second differences divide rounding noise by eps^2
        assert float((o - r).abs().max() / r.abs().max().clamp_min(1.0)) < tol, name
    ref = torch.autograd.grad(sum((o * u).sum() for o, u in zip(ref_out, ups)), params + [inv_s])
    names = [f"plane{i}" for i in range(3)] + [f"line{i}" for i in range(3)] + ["W1", "b1", "W2", "b2", "inv_s"]
    bad = []
    for name, a, b in zip(names, got, ref):
        assert a.shape == b.shape, name
        scale = float(b.abs().max()) + 1e-20
        err = float((a - b).abs().max()) / scale
        l2 = float((a - b).norm() / (b.norm() + 1e-20))
        # the hessian term's adjoint multiplies fp32 rounding noise of the recomputed taps by 1 / eps^2: its share is bounded by g_nh's scale
        if err > 2e-3 or l2 > 2e-3:
            bad.append((name, err, l2))
    assert not bad, bad


def test_sdf_alpha_bwd_single_upstream_terms(dev):
    """Each upstream gradient alone (the others absent = NULL pointers at the ABI), against the composition."""
    from tensoflow_amd import ops
    from tensoflow_amd.autograd import sdf_alpha_composed
    n = 777
    planes, lines, W1, b1, W2, b2, pts, level, dists, dirs, inv_s, ups, units = _setup(dev, n, seed=3)
    packed = ops.VmPacked([p.detach() for p in planes], [l.detach() for l in lines], 3)
    with torch.no_grad():
        alpha, grad, feat, sdf, nh, taps = ops.sdf_alpha(packed, W1, b1, W2, b2, pts, level, dists, dirs, AABB, units, float(inv_s), 0.5, want_taps=True)
    params = planes + lines + [W1, b1, W2, b2]
    for k, name in enumerate(("alpha", "grad", "feat", "sdf", "nh")):
        kw = {"g_" + name: ups[k]}
        gp, g_w1, g_b1, g_w2, g_b2, g_inv = ops.sdf_alpha_bwd(packed, W1.detach(), b1.detach(), W2.detach(), b2.detach(), pts, level, dists, dirs,
                                                              AABB, units, float(inv_s), 0.5, sdf, taps, **kw)
        gplanes, glines = packed.unpack_grad(gp, planes, lines)
        ref_out = sdf_alpha_composed(planes, lines, W1, b1, W2, b2, pts, level, dists, dirs, inv_s, 0.5, AABB, units, 3)
        ref = torch.autograd.grad((ref_out[k] * ups[k]).sum(), params + [inv_s], allow_unused=True)
        got = list(gplanes) + list(glines) + [g_w1, g_b1, g_w2, g_b2, g_inv.reshape(())]
        for a, b, pn in zip(got, ref, range(11)):
            b = torch.zeros_like(a) if b is None else b
            # (a term that cancels exactly in the mathematics -- e.g. d b2[0] under g_grad alone: the +- taps' adjoints sum to zero --
            # is fp32 noise of size ~1e-4 in either implementation: an absolute floor on the scale)
            scale = max(float(b.abs().max()), 0.1)
            assert float((a - b).abs().max()) / scale < 2e-3, (name, pn, float((a - b).abs().max()) / scale)
