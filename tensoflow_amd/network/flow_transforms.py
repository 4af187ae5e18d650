"""The coupling transforms of TensoFlow for the configurations the fused HIP kernels do not instantiate (`flow='pwlinear'`, `'realnvp'`,
any `n_bins != 10`): a differentiable composition of device-resident torch ops around the HIP dense-layer kernels -- correct, not fast; no
shipped config selects them (the fused kernels serve the reference default `flow='pwquad', n_bins=10`).

Reference: network/flow.py:166-168 (modified_softmax), :174-312 (ElementWisePWLinearTransform), :314-525
(ElementWisePWQuadraticTransform), :527-547 (ElementWiseAffineTransform), :123-144 (the sigmoid output cell), :9-24 (Gaussian prior).
Every coupling function takes the coordinates being moved, x or y [M, k], and the net's outputs [M, k, B] and
returns (moved coordinates [M, k], log-Jacobian [M, 1], bin index [M, k] int64).  Directions: `*_density` is the reference's `flow_inv`
(data -> latent, used by TensoFlow.forward), `*_sample` its `flow` (latent -> data, used by TensoFlow.sample)."""
import torch


def _last_not_above(edges, x, first_is_free, lift):
    """Index of the last entry of the non-decreasing `edges` [M,k,E] that is <= x [M,k] -- as the reference finds it: arg-max (FIRST
    occurrence) of [eps, (edges <= x) * (edges + lift)] (flow.py:355-366: lift 0 over the upper bin edges; :443-457: lift 1 over the
    cumulative integrals, result shifted by one).  Ties between equal edges (rounded cumulative sums) therefore resolve to the
    earlier index, exactly as there."""
    eps = torch.finfo(edges.dtype).eps
    found = torch.where(edges > x.unsqueeze(-1), torch.zeros_like(edges), edges + lift)
    head = torch.full_like(edges[..., :1], eps)
    idx = torch.argmax(torch.cat([head, found], -1), dim=-1)
    return idx - 1 if first_is_free else idx


def _quad_tables(wv, clamp_w):
    """wv [M,k,2b+1] -> bin widths w [M,k,b], left edges [M,k,b+1] (cumulative widths, 0 first), knot heights v [M,k,b+1] (normalised so
    that the piecewise-linear density integrates to 1), cumulative integrals vw [M,k,b+1] (0 first)."""
    nv = (wv.shape[-1] + 1) // 2
    v_t, w_t = wv[..., :nv], wv[..., nv:]
    w = torch.exp(w_t)
    if clamp_w:
        w = w.clamp_min(1e-6)
    csum = torch.cumsum(w, -1)
    total = csum[..., -1:]
    w = w / total
    if clamp_w:
        w = w.clamp_min(1e-6)
    csum = csum / total
    left = torch.cat([torch.zeros_like(csum[..., :1]), csum], -1)
    ev = torch.exp(v_t)
    v = (ev / (((ev[..., :-1] + ev[..., 1:]) / 2 * w).sum(-1, keepdim=True))).clamp_min(1e-6)
    vw = torch.cat([torch.zeros_like(v[..., :1]), torch.cumsum((v[..., :-1] + v[..., 1:]) / 2 * w, -1)], -1)
    return w, csum, left, v, vw


def _pick(t, idx):
    return torch.gather(t, -1, idx.unsqueeze(-1)).squeeze(-1)


def pwquad_density(x, wv):
    """flow.py:332-413: the spline itself, data -> latent; log-Jacobian = + log of the interpolated knot height."""
    w, csum, left, v, vw = _quad_tables(wv, clamp_w=True)
    b = w.shape[-1]
    m = _last_not_above(csum, x, first_is_free=False, lift=0.0).clamp(0, b - 1)
    wm, vm, vm1 = _pick(w, m), _pick(v, m), _pick(v, m + 1)
    al = ((x - _pick(left, m)) / wm).clamp(0, 1)
    out = (al ** 2) / 2 * ((vm1 - vm) * wm) + al * vm * wm + _pick(vw, m)
    eps = torch.finfo(out.dtype).eps
    out = out.clamp(eps, 1.0 - eps)
    logj = torch.log(torch.lerp(vm, vm1, al)).sum(-1, keepdim=True)
    return out, logj, m


def pwquad_sample(y, wv):
    """flow.py:415-525: the inverse spline, latent -> data: bin from the cumulative integrals, closed-form root of the bin's quadratic."""
    w, csum, left, v, vw = _quad_tables(wv, clamp_w=False)
    b = w.shape[-1]
    e = _last_not_above(vw, y, first_is_free=True, lift=1.0).clamp(0, b - 1)
    we, ve, ve1 = _pick(w, e), _pick(v, e), _pick(v, e + 1)
    qa = (ve1 - ve) * we
    qb = ve * we
    qc = _pick(vw, e) - y
    eps = torch.finfo(qa.dtype).eps
    qa = torch.where(qa.abs() < eps, torch.full_like(qa, eps), qa)
    disc = (qb ** 2 - 2 * qa * qc).clamp_min(0)
    r1, r2 = (-qb - torch.sqrt(disc)) / qa, (-qb + torch.sqrt(disc)) / qa
    s = torch.where((r1 >= 0) & (r1 < 1), r1, r2).clamp(eps, 1.0 - eps)
    x = (we * s + _pick(left, e)).clamp(eps, 1.0 - eps)
    logj = -torch.log(torch.lerp(ve, ve1, s)).sum(-1, keepdim=True)
    return x, logj, e


def _linear_tables(q_t):
    """q_tilde [M,k,b] -> slopes q [M,k,b] (b * softmax, floored at 1e-6: equal-width bins), integrals strictly left of every bin [M,k,b]."""
    width = 1.0 / q_t.shape[-1]                      # (the reference's constants: 1 / (1 / b) and products with 1 / b, not quotients by b)
    q = (1.0 / width) * torch.softmax(q_t, dim=-1).clamp_min(1e-6)
    cum = torch.cumsum(q, -1) * width
    return q, torch.cat([torch.zeros_like(cum[..., :1]), cum[..., :-1]], -1)


def pwlinear_density(x, q_t):
    """flow.py:193-252: piecewise-linear CDF over b equal bins, data -> latent; log-Jacobian = + log slope."""
    q, left_int = _linear_tables(q_t)
    b = q.shape[-1]
    m = torch.clamp(torch.floor(b * x), 0, b - 1).to(torch.long)
    slope = _pick(q, m)
    out = (x - m * (1.0 / b)) * slope + _pick(left_int, m)      # (1.0 / b: the reference's bin width)
    eps = torch.finfo(out.dtype).eps
    out = out.clamp(eps, 1.0 - eps)
    return out, torch.log(slope).sum(-1, keepdim=True), m


def pwlinear_sample(y, q_t):
    """flow.py:254-312: its inverse, latent -> data: the bin is the one whose left integral is the largest not above y (arg-min of
    y - left over the non-negative differences, first occurrence)."""
    q, left_int = _linear_tables(q_t)
    b = q.shape[-1]
    gap = (y.unsqueeze(-1) - left_int).detach()
    gap = torch.where(gap < 0, torch.full_like(gap, 2.0), gap)
    e = torch.clamp(torch.argmin(gap, dim=-1), 0, b - 1)
    slope = _pick(q, e)
    x = (y - _pick(left_int, e)) / slope + e * (1.0 / b)
    eps = torch.finfo(x.dtype).eps
    x = x.clamp(eps, 1.0 - eps)
    return x, -torch.log(slope).sum(-1, keepdim=True), e


def affine_sample(y, st):
    """ElementWiseAffineTransform.flow (flow.py:528-537), latent -> data: x = exp(s) y + t; st [M,k,2] = (s, t)."""
    es = torch.exp(st[..., 0])
    return es * y + st[..., 1], torch.log(es.clamp_min(1e-6)).sum(-1, keepdim=True), torch.zeros_like(y, dtype=torch.long)


def affine_density(x, st):
    """ElementWiseAffineTransform.flow_inv (flow.py:539-547), data -> latent: exp(-s) (x - t)."""
    es = torch.exp(-st[..., 0])
    return es * (x - st[..., 1]), torch.log(es.clamp_min(1e-6)).sum(-1, keepdim=True), torch.zeros_like(x, dtype=torch.long)


def sigmoid_cell_sample(x):
    """InvertibleAnalyticSigmoid.flow (flow.py:130-136): the output cell of 'realnvp', R^d -> (0,1)^d.  -> (y, log-Jacobian [M,1])."""
    y = torch.sigmoid(x).clamp(1e-6, 1 - 1e-6)
    return y, torch.log((y * (1 - y)).clamp_min(1e-6)).sum(-1, keepdim=True)


def sigmoid_cell_density(x):
    """InvertibleAnalyticSigmoid.flow_inv (flow.py:138-144; inverse_sigmoid :123-124)."""
    return torch.log((x / (1 - x)).clamp_min(1e-6)), -torch.log((x * (1 - x)).clamp_min(1e-6)).sum(-1, keepdim=True)


def gaussian_log_prob(x):
    """FactorizedGaussianSampler.log_prob (flow.py:18-19): sum over the coordinates of Normal(0, 1).log_prob."""
    return (-0.5 * x * x - 0.9189385332046727).sum(-1, keepdim=True)          # log sqrt(2 pi)


TRANSFORMS = {"pwquad": (pwquad_sample, pwquad_density, lambda n: 2 * n + 1), "pwlinear": (pwlinear_sample, pwlinear_density, lambda n: n),
              "realnvp": (affine_sample, affine_density, lambda n: 2)}
