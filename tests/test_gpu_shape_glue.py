"""tf_shape_glue_* (round 6): the element-wise algebra of ShapeShadingNetwork.forward in the training direction (network/fields.py:448-567)
as two differentiable launches, against the torch composition it replaces -- values and every gradient, on inputs that visit the edges
(degenerate normals, back-facing views, roughness / NoV outside the LUT's interior, occlusion logits outside [0, 1], colours in the
linear toe of the sRGB curve and beyond 1).  The reference-run gradient goldens of the renderer tests cover the same code end to end."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _composition(normals, view, mat, dl, dr, il, occ_raw, lut):
    """ShapeShadingNetwork._composed's algebra, as it was written in torch (the path kept for inter_results / human_light)."""
    from tensoflow_amd.encodings import linear_to_srgb
    n = F.normalize(normals, dim=-1)
    bad = (n[:, :2].sum(-1) == 0.0)[:, None]
    n = torch.where(bad, torch.tensor([0.0, 1e-6, 1.0], device=n.device, dtype=n.dtype), n)
    v = F.normalize(view, dim=-1)
    nov = (n * v).sum(-1, keepdim=True)
    refl = nov * n * 2 - v
    albedo, rough, metal = mat[..., :3] * 0.77 + 0.03, mat[..., 3:4] * 0.9 + 0.09, mat[..., 4:]
    occ_prob = occ_raw * 0.5 + 0.5
    occ = occ_prob.clamp(0, 1)
    light = il * occ + dr * (1 - occ)
    uv = torch.cat([nov.clamp(0, 1), rough.clamp(0, 1)], -1)
    l4 = lut.reshape(lut.shape[-3], lut.shape[-2], 2).permute(2, 0, 1)[None]
    fg = F.grid_sample(l4, (uv * 2 - 1)[None, :, None, :], mode="bilinear", padding_mode="border", align_corners=False)[0, :, :, 0].t()
    sref = (0.04 * (1 - metal) + metal * albedo) * fg[:, 0:1] + fg[:, 1:2]
    color = linear_to_srgb((1 - metal) * albedo * dl + sref * light).clamp(0.0, 1.0)
    return n, v, nov, refl, rough, color, occ_prob


def _inputs(dev, n=20000, seed=0):
    g = torch.Generator().manual_seed(seed)
    normals = torch.randn(n, 3, generator=g) * torch.exp(torch.randn(n, 1, generator=g))
    normals[:50] = 0.0                                    # F.normalize's eps branch, then the (0, 1e-6, 1) patch
    normals[50:100, :2] = 0.0                             # n.x + n.y == 0: patched
    normals[100:150, 1] = -normals[100:150, 0]            # n.x + n.y == 0 by cancellation
    view = torch.randn(n, 3, generator=g)
    mat = torch.rand(n, 5, generator=g)
    mat[150:300, 3] = torch.tensor([0.0, 1.0]).repeat(75)  # roughness 0.09 / 0.99: first / last LUT rows
    mat[300:400, 3] = 1.2                                  # beyond the sigmoid's range: roughness > 1 exercises the clamp mask
    dl, dr, il = (torch.rand(n, 3, generator=g) * s for s in (1.5, 2.0, 1.0))
    dl[400:600] *= 1e-3                                   # the linear toe of the sRGB curve
    dr[600:700] *= 20.0                                   # saturated colours (clamp to 1)
    occ_raw = torch.randn(n, 1, generator=g) * 1.5          # occ_prob outside [0, 1] on a good part of the rows
    lut = torch.rand(1, 64, 48, 2, generator=g)
    return [t.to(dev) for t in (normals, view, mat, dl, dr, il, occ_raw, lut)]


def test_shape_glue_matches_the_torch_composition_values_and_gradients():
    from tensoflow_amd.autograd import ShapeGluePostFn, ShapeGluePreFn
    dev = torch.device("cuda:0")
    normals, view, mat, dl, dr, il, occ_raw, lut = _inputs(dev)
    leaves = [normals, mat, dl, dr, il, occ_raw]
    w = [torch.randn_like(t) for t in (normals, view[:, :1], normals, view[:, :1], dl, occ_raw)]      # weights of n_u, nov, refl, rough, color, occ_prob

    def run(fused, dt=torch.float32):
        xs = [t.clone().to(dt).requires_grad_(True) for t in leaves]
        nn_, m_, dl_, dr_, il_, oc_ = xs
        if fused:
            nu, vu, nov, refl, rough, _ = ShapeGluePreFn.apply(nn_, view, m_)
            color, occ_prob = ShapeGluePostFn.apply(m_, nov, dl_, dr_, il_, oc_, lut)
        else:
            nu, vu, nov, refl, rough, color, occ_prob = _composition(nn_, view.to(dt), m_, dl_, dr_, il_, oc_, lut.to(dt))
        outs = (nu, nov, refl, rough, color, occ_prob)
        loss = sum((o * wi.to(dt)).sum() for o, wi in zip(outs, w))
        loss.backward()
        return [o.detach() for o in outs] + [vu.detach()], [x.grad for x in xs]

    # three evaluations of one function: the two launches, the torch composition in fp32, the torch composition in fp64 (the truth).  The
    # LUT here is white noise (slope ~ its width per unit of NoV) and the sRGB toe has slope 12.92: fp32 rounding of NoV / the linear colour
    # shows as ~1e-5 in the colour and ~1e-3 of a gradient row in EITHER fp32 evaluation, so the bar is the composition's own distance
    # from the truth: the fused form may be at most twice as far (+ a floor).
    (o_f, g_f), (o_r, g_r), (o_t, g_t) = run(True), run(False), run(False, torch.float64)
    for name, a, b, t in zip(("normals_u", "nov", "reflective", "roughness", "color", "occ_prob", "view_u"), o_f, o_r, o_t):
        e_f, e_r = float((a - t).abs().max()), float((b - t).abs().max())
        assert e_f <= 2.0 * e_r + 1e-6, (name, e_f, e_r)
    for name, a, b, t in zip(("normals", "mat", "diffuse_light", "direct_light", "indirect_light", "occ_raw"), g_f, g_r, g_t):
        scale = t.abs().amax(-1, keepdim=True).clamp_min(1e-3 * float(t.abs().max()))       # per row: tiny raw normals give 1 / |x| gradients
        e_f, e_r = float(((a - t).abs() / scale).max()), float(((b - t).abs() / scale).max())
        print(f"grad {name}: fused {e_f:.2e}, torch fp32 {e_r:.2e} from the fp64 composition")
        assert e_f <= 2.0 * e_r + 1e-5, (name, e_f, e_r)
        # ... and the same branch decisions everywhere: where the fp64 gradient row is exactly zero (patched normals, clamped LUT
        # coordinates, saturated colours) the fused one is zero too, except on rows that sit within rounding of a branch point
        zf, zt = (a == 0).all(-1), (t == 0).all(-1)
        assert float((zf != zt).float().mean()) < 2e-3, name
    # a patched row receives no gradient; a saturated colour passes none on
    assert float(g_f[0][50:150].abs().max()) == 0.0


def test_shape_glue_missing_gradients_and_empty_batch():
    """Only the colour is differentiated (occ_prob unused; roughness / reflective without consumers): the adjoints take NULL for the
    gradients that did not arrive; n = 0 returns empty tensors."""
    from tensoflow_amd.autograd import ShapeGluePostFn, ShapeGluePreFn
    dev = torch.device("cuda:0")
    normals, view, mat, dl, dr, il, occ_raw, lut = _inputs(dev, n=513, seed=3)
    m = mat.clone().requires_grad_(True)
    nn_ = normals.clone().requires_grad_(True)
    nu, vu, nov, refl, rough, _ = ShapeGluePreFn.apply(nn_, view, m)
    color, occ_prob = ShapeGluePostFn.apply(m, nov, dl, dr, il, occ_raw, lut)
    color.sum().backward()
    assert torch.isfinite(m.grad).all() and torch.isfinite(nn_.grad).all() and float(m.grad.abs().sum()) > 0
    e3, e5, e1 = torch.zeros(0, 3, device=dev), torch.zeros(0, 5, device=dev), torch.zeros(0, 1, device=dev)
    out = ShapeGluePreFn.apply(e3, e3, e5)
    assert [tuple(t.shape) for t in out[:5]] == [(0, 3), (0, 3), (0, 1), (0, 3), (0, 1)]
    c, o = ShapeGluePostFn.apply(e5, e1, e3, e3, e3, e1, lut)
    assert tuple(c.shape) == (0, 3) and tuple(o.shape) == (0, 1)


def test_shape_glue_mip_coordinate_equals_envlight_get_mip():
    """The specular-stack coordinate the prelude emits with the roughness is EnvLight.get_mip(roughness).clamp(0, n - 1) (network/light.py:
    72-80, :101) -- value and gradient wrt mat, on roughness values that visit both linear pieces, their joint and both clamps."""
    from tensoflow_amd.autograd import ShapeGluePreFn
    from tensoflow_amd.network.light import EnvLight
    dev = torch.device("cuda:0")
    env = EnvLight(trainable=False, max_res=128, device=dev)
    env.build_mips()
    n_lv = len(env.specular)
    normals, view, mat, *_ = _inputs(dev, n=4096, seed=5)
    mat[:, 3] = torch.linspace(-0.2, 1.2, 4096, device=dev)          # roughness 0.09 - 0.18 .. 0.09 + 1.08
    mat[7, 3] = (env.max_roughness - 0.09) / 0.9                       # the joint of the two pieces
    m1 = mat.clone().requires_grad_(True)
    *_, rough, mip = ShapeGluePreFn.apply(normals, view, m1, (env.min_roughness, env.max_roughness, n_lv))
    w = torch.randn(4096, device=dev)
    (mip * w).sum().backward()
    m2 = mat.clone().requires_grad_(True)
    ref = env.get_mip((m2[:, 3] * 0.9 + 0.09)).clamp(0, n_lv - 1)
    (ref * w).sum().backward()
    assert float((mip - ref).abs().max()) < 1e-5
    same = (m1.grad[:, 3] - m2.grad[:, 3]).abs() <= 1e-4 * m2.grad[:, 3].abs().max()
    assert float(same.float().mean()) > 0.998 and float(m1.grad[:, [0, 1, 2, 4]].abs().max()) == 0.0      # (rows within rounding of a branch point may differ)


def test_normalize3_values_and_gradients_against_torch():
    """tf_normalize3_*: F.normalize of [n,3] rows with the eikonal residual, and of the opacity blend with a constant, against torch in fp64
    (rows of length 0, below the eps, ~1 and large)."""
    from tensoflow_amd.autograd import Normalize3Fn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    n = 10000
    x = (torch.randn(n, 3, generator=g) * torch.exp(2 * torch.randn(n, 1, generator=g))).to(dev)
    x[:10] = 0.0
    x[10:20] *= 1e-14
    acc = torch.rand(n, 1, generator=g).to(dev)
    acc[20:30] = 0.0
    wy, we = torch.randn(n, 3, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    c = (0.0, 0.0, 1.0)
    for blend in (False, True):
        xs = x.clone().requires_grad_(True)
        a_ = acc.clone().requires_grad_(True) if blend else None
        y, err = Normalize3Fn.apply(xs, a_, c if blend else None, not blend)
        ((y * wy).sum() + ((err * we).sum() if not blend else 0.0)).backward()
        xt = x.double().clone().requires_grad_(True)
        at = acc.double().clone().requires_grad_(True)
        xb = xt * at + (1 - at) * torch.tensor(c, device=dev, dtype=torch.float64) if blend else xt
        yt = F.normalize(xb, dim=-1)
        et = (torch.linalg.norm(xb, ord=2, dim=-1) - 1.0) ** 2
        ((yt * wy.double()).sum() + ((et * we.double()).sum() if not blend else 0.0)).backward()
        assert float((y - yt).abs().max()) < 1e-6, blend
        if not blend:
            assert float(((err - et).abs() / et.abs().clamp_min(1.0)).max()) < 1e-5
        big = xt.grad.abs().amax(-1) < 1e9              # (rows below the eps have gradients of 1e12: compared relatively)
        rel = (xs.grad - xt.grad).abs().amax(-1) / xt.grad.abs().amax(-1).clamp_min(1e-6)
        assert float(rel[big].max()) < 1e-4 and float(rel.max()) < 1e-3, (blend, float(rel[big].max()), float(rel.max()))
        if blend:
            assert float(((a_.grad - at.grad).abs() / at.grad.abs().clamp_min(1.0)).max()) < 1e-4
