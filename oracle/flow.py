"""Oracle: TensoFlow conditional 2-D normalising flow, 'pwquad' variant (TEST INFRASTRUCTURE).

Follows network/flow.py:
  SphereSampler :52-90, Reshift :146-164, modified_softmax :166-168,
  ElementWisePWQuadraticTransform.flow_inv :332-413 / .flow :415-525,
  Block :549-641, TensoFlow.tenso_feature :709-744, .flow :766-780, .flow_inv :782-799,
  .forward :801-831, .sample :833-855.

State-dict keys (relative to the TensoFlow module): nis_plane.{0,1,2}, nis_line.{0,1,2},
nis_mat.{0,2}.{weight,bias}, flows.{0,1}.nn.{1,3,5,7}.{weight,bias}.
Eval mode only (SphereSampler's azimuth jitter is injected by the caller as `jitter`).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .encodings import posenc
from .vm_field import vm_feature

EPS32 = float(torch.finfo(torch.float32).eps)


# ----------------------------------------------------------------------------- prior
def sphere_latent(sn):
    """Upper-hemisphere Fibonacci set, begin elevation 1 degree (flow.py:62-76) -> [sn,2]."""
    ratio = (1 + 90) / 180
    num_points = int(sn // (1 - ratio))
    g = (np.sqrt(5) - 1.0) / 2.0
    phis, thetas = [], []
    for n in range(num_points - sn, num_points):
        z = 2.0 * n / num_points - 1.0
        phis.append(2 * np.pi * n * g % (2 * np.pi))
        thetas.append(np.arcsin(z))
    phi = (torch.tensor(phis, dtype=torch.float32) / (2 * np.pi)).to(torch.get_default_dtype())
    theta = (torch.tensor(thetas, dtype=torch.float32) / (0.5 * np.pi)).to(torch.get_default_dtype())
    return torch.stack([phi, theta], -1)


def sphere_prior(pn, sn, jitter=None):
    """-> x [pn,sn,2], logj [pn,sn,1]   (flow.py:82-90); jitter [pn,sn,1] in [0,1) or None."""
    x = sphere_latent(sn).expand(pn, sn, 2)
    if jitter is not None:
        x = torch.cat([(x[..., :1] + jitter) % 1, x[..., 1:]], -1)
    x = x.clamp(1e-6, 1 - 1e-6)
    logj = -torch.log(torch.cos(x[..., 1:] * (0.5 * np.pi)))
    return x, logj


def prior_log_prob(z):
    return torch.log(torch.cos(z[..., 1:] * (0.5 * np.pi)))


# ------------------------------------------------------------------- piecewise quadratic
def _pwquad_tables(wv, clamp_w):
    """wv [M, 2b+1] -> (w [M,b], wsum_shift [M,b+1], v [M,b+1], vw [M,b+1])."""
    nb1 = int(math.ceil(wv.shape[-1] / 2))
    v_t, w_t = wv[:, :nb1], wv[:, nb1:]
    w = torch.exp(w_t)
    if clamp_w:
        w = w.clamp_min(1e-6)
    wsum = torch.cumsum(w, -1)
    wn = wsum[:, -1:]
    w = w / wn
    if clamp_w:
        w = w.clamp_min(1e-6)
    wsum = wsum / wn
    wsum_shift = torch.cat([torch.zeros_like(wsum[:, :1]), wsum], -1)
    ev = torch.exp(v_t)
    v = ev / ((ev[:, :-1] + ev[:, 1:]) / 2 * w).sum(-1, keepdim=True)
    v = v.clamp_min(1e-6)
    vw = torch.cat([torch.zeros_like(v[:, :1]), torch.cumsum((v[:, :-1] + v[:, 1:]) / 2 * w, -1)], -1)
    return w, wsum_shift, v, vw


def _last_leq(edges_hi, val, offset):
    """index of the last entry of `edges_hi` (increasing, [M,b]) that is <= val, via the
    reference's arg-max trick (flow.py:355-366 / :443-453); returns long [M]."""
    finder = torch.where(edges_hi > val[:, None], torch.zeros_like(edges_hi), torch.ones_like(edges_hi))
    probe = torch.cat([torch.full_like(edges_hi[:, :1], EPS32), finder * (edges_hi + offset)], -1)
    return torch.argmax(probe, -1)


def pwquad_inverse(y, wv):
    """Sampling direction (`.flow`, flow.py:415-525): y [M] in (0,1), wv [M,2b+1]
    -> x [M], logj [M], bin [M] long."""
    w, wss, v, vw = _pwquad_tables(wv, clamp_w=False)
    nb = w.shape[-1]
    mx = _last_leq(vw, y, 1.0) - 1          # probe over the b+1 entries of vw (incl. leading 0)
    e = mx.clamp(0, nb - 1)
    g = lambda t, i: torch.gather(t, -1, i[:, None])[:, 0]
    ve, ve1, we = g(v, e), g(v, e + 1), g(w, e)
    a = (ve1 - ve) * we
    b = ve * we
    c = g(vw, e) - y
    a = torch.where(a.abs() < EPS32, torch.full_like(a, EPS32), a)
    d = (b ** 2 - 2 * a * c).clamp_min(0)
    s1 = (-b - torch.sqrt(d)) / a
    s2 = (-b + torch.sqrt(d)) / a
    sol = torch.where((s1 >= 0) & (s1 < 1), s1, s2).clamp(EPS32, 1 - EPS32)
    x = (we * sol + g(wss, e)).clamp(EPS32, 1 - EPS32)
    logj = -torch.log(torch.lerp(ve, ve1, sol))
    return x, logj, e


def pwquad_forward(x, wv):
    """Density direction (`.flow_inv`, flow.py:332-413): x [M] -> out [M], logj [M], bin."""
    w, wss, v, vw = _pwquad_tables(wv, clamp_w=True)
    nb = w.shape[-1]
    mx = _last_leq(wss[:, 1:], x, 0.0).clamp(0, nb - 1)
    g = lambda t, i: torch.gather(t, -1, i[:, None])[:, 0]
    vm, vm1, wm = g(v, mx), g(v, mx + 1), g(w, mx)
    al = ((x - g(wss, mx)) / wm).clamp(0, 1)
    out = (al ** 2) / 2 * ((vm1 - vm) * wm) + al * vm * wm + g(vw, mx)
    out = out.clamp(EPS32, 1 - EPS32)
    logj = torch.log(torch.lerp(vm, vm1, al))
    return out, logj, mx


# ------------------------------------------------------------------------------- blocks
def _block_net(sd, pfx, y_keep, cond):
    h = torch.cat([posenc(y_keep, 3), cond], -1) * 2.0 - 1.0      # Reshift(2,-1)
    for li in (1, 3, 5):
        h = F.leaky_relu(F.linear(h, sd[f"{pfx}.nn.{li}.weight"], sd[f"{pfx}.nn.{li}.bias"]))
    return F.linear(h, sd[f"{pfx}.nn.7.weight"], sd[f"{pfx}.nn.7.bias"])


def block_apply(sd, pfx, keep, y, logj, cond, inverse):
    """One coupling block on y [M,2]; `keep` = index of the coordinate that is kept."""
    move = 1 - keep
    wv = _block_net(sd, pfx, y[:, keep:keep + 1], cond)
    fn = pwquad_forward if inverse else pwquad_inverse
    t, lj, bins = fn(y[:, move], wv)
    out = torch.zeros_like(y)
    out[:, keep] = y[:, keep]
    out[:, move] = t
    return out, logj + lj[:, None], bins


# ---------------------------------------------------------------------------- TensoFlow
def flow_condition(sd, pts, view_angles, roughness, aabb, pfx="", n_levels=3, ablate=(False, False)):
    """[pn,37] = [nis feature 16 | embed3(view_angles) 14 | 0*embed3(roughness) 7]."""
    planes = [sd[f"{pfx}nis_plane.{i}"] for i in range(3)]
    lines = [sd[f"{pfx}nis_line.{i}"] for i in range(3)]
    feat = vm_feature(planes, lines, pts, aabb, None, n_levels)
    h = torch.cat([feat, posenc(pts, 3)], -1)
    h = F.softplus(F.linear(h, sd[f"{pfx}nis_mat.0.weight"], sd[f"{pfx}nis_mat.0.bias"]), beta=100)
    h = F.linear(h, sd[f"{pfx}nis_mat.2.weight"], sd[f"{pfx}nis_mat.2.bias"])
    refl = posenc(view_angles, 3)
    if ablate[0]:                      # disable_tensorial (flow.py:807-808, :838-839)
        h = torch.zeros_like(h)
    if ablate[1]:                      # disable_reflected (flow.py:811-812, :842-843)
        refl = torch.zeros_like(refl)
    return torch.cat([h, refl, torch.zeros_like(posenc(roughness, 3))], -1)


def flow_sample(sd, pts, view_angles, roughness, sn, aabb, pfx="", jitter=None, return_bins=False, ablate=(False, False)):
    """TensoFlow.sample(..., return_jacobian=True) -> angles [pn,sn,2], logj [pn,sn,1]."""
    pn = pts.shape[0]
    x, logj = sphere_prior(pn, sn, jitter)
    cond = flow_condition(sd, pts, view_angles, roughness, aabb, pfx, ablate=ablate)
    cond = cond[:, None, :].expand(pn, sn, cond.shape[-1]).reshape(pn * sn, -1)
    y, lj = x.reshape(-1, 2), logj.reshape(-1, 1)
    y, lj, b0 = block_apply(sd, f"{pfx}flows.0", 0, y, lj, cond, inverse=False)
    y, lj, b1 = block_apply(sd, f"{pfx}flows.1", 1, y, lj, cond, inverse=False)
    out = (y.reshape(pn, sn, 2), lj.reshape(pn, sn, 1))
    if return_bins:
        return out + (b0.reshape(pn, sn), b1.reshape(pn, sn))
    return out


def flow_logq(sd, pts, view_angles, roughness, x, aabb, pfx="", rays_id=None, return_bins=False):
    """TensoFlow.forward(..., return_jacobian=True) -> z, logqx.
    x is [pn,sn,2] (rays_id None) or [M,2] with rays_id [M]."""
    x = x.clamp(1e-6, 1 - 1e-6)
    cond = flow_condition(sd, pts, view_angles, roughness, aabb, pfx)
    shape = x.shape[:-1]
    if rays_id is not None:
        cond = cond[rays_id]
    else:
        cond = cond[:, None, :].expand(shape[0], shape[1], cond.shape[-1]).reshape(-1, cond.shape[-1])
    y = x.reshape(-1, 2)
    lj = torch.zeros(y.shape[0], 1)
    y, lj, b1 = block_apply(sd, f"{pfx}flows.1", 1, y, lj, cond, inverse=True)
    y, lj, b0 = block_apply(sd, f"{pfx}flows.0", 0, y, lj, cond, inverse=True)
    z = y.reshape(*shape, 2)
    logq = lj.reshape(*shape, 1) + prior_log_prob(z)
    if return_bins:
        return z, logq, b0.reshape(shape), b1.reshape(shape)
    return z, logq
