"""Oracle flow restatement vs golden vectors produced by the imported reference."""
import torch

from conftest import AABB, rel_err
from oracle import flow as oflow


def test_pwquad(golden):
    g = golden("pwquad")
    x, lj, bins = oflow.pwquad_inverse(g["y"], g["wv"])
    assert rel_err(x, g["inv_x"]) < 1e-6
    assert rel_err(lj, g["inv_logj"]) < 1e-5
    out, lji, bins_f = oflow.pwquad_forward(g["y"], g["wv"])
    assert rel_err(out, g["fwd_out"]) < 1e-6
    assert rel_err(lji, g["fwd_logj"]) < 1e-5
    assert bins.min() >= 0 and bins.max() <= 9 and bins_f.min() >= 0 and bins_f.max() <= 9
    # round trip through the oracle itself: forward(inverse(y)) == y
    back, ljb, _ = oflow.pwquad_forward(x, g["wv"])
    ok = (g["wv"][:, 11:].abs().max(-1).values < 8)          # skip the deliberately degenerate rows
    # (the reference's closed-form root loses digits when the quadratic coefficient is tiny,
    #  so this is a statement about the bulk, not the worst row)
    assert float(torch.quantile((back - g["y"]).abs()[ok], 0.99)) < 1e-4


def test_latent(golden):
    g = golden("tensoflow_r32")
    for sn in (8, 32, 128):
        x, lj = oflow.sphere_prior(1, sn)
        assert rel_err(x[0], g[f"latent_{sn}"]) < 1e-7
        assert rel_err(lj[0], g[f"latent_logj_{sn}"]) < 1e-6


def test_flow_sample_and_logq(golden):
    g = golden("tensoflow_r32")
    for sn in (8, 32, 128):
        ang, logj = oflow.flow_sample(g.sd, g["pts"], g["view_angles"], g["roughness"], sn, AABB)
        assert rel_err(ang, g[f"angles_{sn}"]) < 2e-6
        assert rel_err(logj, g[f"logj_{sn}"]) < 2e-5
        z, logq = oflow.flow_logq(g.sd, g["pts"], g["view_angles"], g["roughness"], g[f"angles_{sn}"], AABB)
        assert rel_err(z, g[f"z_{sn}"]) < 2e-6
        assert rel_err(logq, g[f"logq_{sn}"]) < 2e-5
    z, logq = oflow.flow_logq(g.sd, g["pts"], g["view_angles"], g["roughness"], g["x_rand"], AABB)
    assert rel_err(z, g["z_rand"]) < 2e-6 and rel_err(logq, g["logq_rand"]) < 2e-5
    z, logq = oflow.flow_logq(g.sd, g["pts"], g["view_angles"], g["roughness"], g["x_rid"], AABB, rays_id=g["rays_id"])
    assert rel_err(z, g["z_rid"]) < 2e-6 and rel_err(logq, g["logq_rid"]) < 2e-5


def test_flow_backward(golden):
    g = golden("tensoflow_r32")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in g.sd.items()}
    z, logq = oflow.flow_logq(sd, g["pts"], g["view_angles"], g["roughness"], g["x_rand"], AABB)
    (-(g["bwd_w"] * logq).mean()).backward()
    for k, ref in g.grad.items():
        assert rel_err(sd[k].grad, ref) < 2e-5, k
