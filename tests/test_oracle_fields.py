"""Oracle (CPU restatement) vs golden vectors produced by the imported reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import AABB, rel_err
from oracle import encodings as oenc
from oracle import texture as otex
from oracle import vm_field as ovm


@pytest.mark.parametrize("tag", ["r32_l1", "r32_l3", "r24x32x40_l3"])
def test_tensosdf_forward_gradient(golden, tag):
    g = golden("tensosdf_" + tag)
    nl = int(g["n_levels"])
    out_none = ovm.sdf_forward(g.sd, g["pts"], None, AABB, nl)
    out_lvl = ovm.sdf_forward(g.sd, g["pts"], g["level"], AABB, nl)
    assert rel_err(out_none, g["out_none"]) < 2e-6
    assert rel_err(out_lvl, g["out_lvl"]) < 2e-6
    grad, nh = ovm.sdf_gradient(g.sd, g["pts"], g["level"], AABB, nl, g["grid_size"], sdf=out_lvl[:, :1], training=True)
    assert rel_err(grad, g["grad_lvl"]) < 1e-5
    assert rel_err(nh, g["normal_hessian"]) < 1e-4     # second differences / eps^2 amplify rounding
    grad0, _ = ovm.sdf_gradient(g.sd, g["pts"], None, AABB, nl, g["grid_size"])
    assert rel_err(grad0, g["grad_none"]) < 1e-5


@pytest.mark.parametrize("tag", ["mr3", "mr2"])
def test_tensosdf_multires_forward_gradient_backward(golden, tag):
    """sdf_multires > 0 (fields.py:66-91: the reference class's default is 3; no shipped config sets it): the positional encoding of
    the contracted (m = 3) or raw (m = 2) point in front of the decoder, against the imported reference."""
    g = golden("tensosdf_" + tag)
    assert g.sd["sdf_mat.0.weight"].shape[1] == 108 + 3 + 6 * int(g["multires"])
    out_none = ovm.sdf_forward(g.sd, g["pts"], None, AABB, 3)
    out_lvl = ovm.sdf_forward(g.sd, g["pts"], g["level"], AABB, 3)
    assert rel_err(out_none, g["out_none"]) < 2e-6 and rel_err(out_lvl, g["out_lvl"]) < 2e-6
    grad, nh = ovm.sdf_gradient(g.sd, g["pts"], g["level"], AABB, 3, g["grid_size"], sdf=out_lvl[:, :1], training=True)
    assert rel_err(grad, g["grad_lvl"]) < 1e-5 and rel_err(nh, g["normal_hessian"]) < 2e-4
    sd = {k: v.clone().requires_grad_(True) for k, v in g.sd.items()}
    (ovm.sdf_forward(sd, g["pts"], g["level"], AABB, 3) * g["bwd_w"]).sum().backward()
    for k, ref in g.grad.items():
        assert rel_err(sd[k].grad, ref) < 1e-5, k


def test_tensosdf_backward(golden):
    g = golden("tensosdf_r32_l3")
    sd = {k: v.clone().requires_grad_(True) for k, v in g.sd.items()}
    out = ovm.sdf_forward(sd, g["pts"], g["level"], AABB, 3)
    (out * g["bwd_w"]).sum().backward()
    for k, ref in g.grad.items():
        assert rel_err(sd[k].grad, ref) < 1e-5, k


def test_texture_matches_grid_sample():
    """independent formulation of the 2-D clamp path: grid_sample(align_corners=False, border)."""
    torch.manual_seed(0)
    tex = torch.randn(1, 12, 20, 5)
    uv = torch.rand(1, 300, 1, 2) * 1.4 - 0.2
    for lvl in range(3):
        mips = otex.box_mips(tex[0], 3)
        got = otex.texture(tex, uv, mip_level_bias=torch.full((1, 300, 1), float(lvl)), boundary_mode="clamp",
                           max_mip_level=2)
        ref = F.grid_sample(mips[lvl].permute(2, 0, 1)[None], uv * 2 - 1, mode="bilinear", padding_mode="border",
                            align_corners=False)
        assert rel_err(got[0, :, 0], ref[0, :, :, 0].T) < 2e-5   # coordinate rounding differs
    # fractional level = lerp of the two neighbours; clamps at both ends
    lv = torch.rand(1, 300, 1) * 5 - 1
    got = otex.texture(tex, uv, mip_level_bias=lv, boundary_mode="clamp", max_mip_level=2)
    per = [otex.texture(tex, uv, mip_level_bias=torch.full((1, 300, 1), float(l)), boundary_mode="clamp",
                        max_mip_level=2) for l in range(3)]
    lc = lv.clamp(0, 2)
    l0 = lc.floor().clamp(max=2)
    f = (lc - l0)[..., None]
    idx0, idx1 = l0.long(), (l0.long() + 1).clamp(max=2)
    stack = torch.stack(per, 0)
    pick = lambda idx: torch.gather(stack, 0, idx[None, ..., None].expand(1, 1, 300, 1, 5))[0]
    assert rel_err(got, pick(idx0) * (1 - f) + pick(idx1) * f) < 1e-6


def test_line_texture_mips():
    tex = torch.arange(8.0).reshape(1, 8, 1, 1)
    m = otex.box_mips(tex[0], 3)
    assert [tuple(t.shape) for t in m] == [(8, 1, 1), (4, 1, 1), (2, 1, 1)]
    assert torch.allclose(m[1][:, 0, 0], torch.tensor([0.5, 2.5, 4.5, 6.5]))
    with pytest.raises(ValueError):
        otex.box_mips(torch.zeros(6, 1, 1), 3)   # 6 -> 3 -> odd


def test_cube_lookup_properties():
    torch.manual_seed(1)
    R = 8
    tex = torch.randn(6, R, R, 3)
    # texel centres return the texel exactly
    for s in range(6):
        iy, ix = torch.meshgrid(torch.arange(R), torch.arange(R), indexing="ij")
        x = (ix.reshape(-1) + 0.5) / R * 2 - 1
        y = (iy.reshape(-1) + 0.5) / R * 2 - 1
        d = otex._face_dir(torch.full_like(ix.reshape(-1), s), x, y)
        got = otex.cube_bilinear(tex, d)
        assert torch.allclose(got, tex[s].reshape(-1, 3), atol=1e-5)
        f, x2, y2 = otex.cube_face_uv(d)
        assert (f == s).all() and torch.allclose(x2, x, atol=1e-6) and torch.allclose(y2, y, atol=1e-6)
    # continuity across face seams: a constant-per-direction field (smooth function of the direction)
    # sampled on both sides of an edge must agree to O(texel)
    d = F.normalize(torch.randn(4000, 3), dim=-1)
    smooth = torch.stack([otex._face_dir(torch.full((R * R,), s), *(
        ((torch.arange(R * R) % R + 0.5) / R * 2 - 1), ((torch.arange(R * R) // R + 0.5) / R * 2 - 1)))
        for s in range(6)]).reshape(6, R, R, 3)
    smooth = F.normalize(smooth, dim=-1)
    got = otex.cube_bilinear(smooth, d)                     # interpolating the direction field itself
    assert float((F.normalize(got, dim=-1) - d).norm(dim=-1).max()) < 0.08
    # scale invariance of the direction
    assert torch.allclose(otex.cube_bilinear(tex, d), otex.cube_bilinear(tex, d * 3.7), atol=1e-5)


def test_encodings(golden):
    g = golden("encodings")
    for nf in (3, 6, 8):
        assert rel_err(oenc.posenc(g["x3"], nf), g[f"posenc{nf}"]) < 1e-6
    assert rel_err(oenc.ide5(g["dirs"], 0.0), g["ide5"]) < 1e-5
    assert rel_err(oenc.linear_to_srgb(g["lin"]), g["srgb"]) < 1e-6
