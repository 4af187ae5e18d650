"""Oracle: small closed-form encodings on the hot path (TEST INFRASTRUCTURE).

Restates (does not import) the reference's
  * sin/cos positional encoding  utils/network_utils.py:5-50 (include input; per frequency
    2^k, k=0..n-1: sin then cos),
  * integrated directional encoding, degree 5 -> 72 dims  utils/ref_utils.py:8-117
    (Ref-NeRF eq. 6-8; (l, m) pairs l = 1,2,4,8,16, m = 0..l),
  * sRGB OETF  utils/raw_utils.py:4-11,
  * weight-norm effective weights of nn.utils.parametrizations.weight_norm
    (network/other_field.py:20-119): W = g * v / ||v||_row.
"""
import math

import numpy as np
import torch


def posenc(x, n_freq):
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def _gen_binom(a, k):
    return float(np.prod(a - np.arange(k))) / math.factorial(k)


def _legendre_coeff(l, m, k):
    return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m)
            * _gen_binom(0.5 * (l + k + m - 1.0), l))


def _sph_coeff(l, m, k):
    return math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * math.pi * math.factorial(l + m))) \
        * _legendre_coeff(l, m, k)


def ide_tables(deg=5):
    """-> (m list [36], l list [36], mat float32 [l_max+1, 36])."""
    ms, ls = [], []
    for i in range(deg):
        l = 2 ** i
        for m in range(l + 1):
            ms.append(m)
            ls.append(l)
    l_max = 2 ** (deg - 1)
    mat = np.zeros((l_max + 1, len(ms)))
    for i, (m, l) in enumerate(zip(ms, ls)):
        for k in range(l - m + 1):
            mat[k, i] = _sph_coeff(l, m, k)
    return ms, ls, mat.astype(np.float32)


_IDE = None


def ide5(xyz, kappa_inv=0.0):
    """xyz [...,3] -> [...,72]; kappa_inv scalar or [...,1]."""
    global _IDE
    if _IDE is None:
        ms, ls, mat = ide_tables(5)
        _IDE = (torch.tensor(ms, dtype=torch.float32), torch.tensor(ls, dtype=torch.float32),
                torch.from_numpy(mat))
    ms, ls, mat = [t.to(xyz.dtype) for t in _IDE]       # fp32 in the reference; follows the input for the fp64 re-evaluation
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    vmz = torch.cat([z ** i for i in range(mat.shape[0])], -1)
    zc = x + 1j * y
    vmxy = torch.cat([zc ** m for m in ms], -1)
    sph = vmxy * torch.matmul(vmz, mat)
    sigma = 0.5 * ls * (ls + 1)
    val = sph * torch.exp(-sigma * kappa_inv)
    return torch.cat([val.real, val.imag], -1)


def linear_to_srgb(lin):
    eps = torch.finfo(torch.float32).eps
    s0 = 323 / 25 * lin
    s1 = (211 * torch.clamp(lin, min=eps) ** (5 / 12) - 11) / 200
    return torch.where(lin <= 0.0031308, s0, s1)


def wn_weight(sd, prefix):
    """Effective weight of a weight-normed Linear stored as
    `<prefix>.parametrizations.weight.original0/1` (g [out,1], v [out,in]); falls back to
    a plain `<prefix>.weight`."""
    k0 = prefix + ".parametrizations.weight.original0"
    if k0 in sd:
        g = sd[k0]
        v = sd[prefix + ".parametrizations.weight.original1"]
        return g * v / v.norm(dim=1, keepdim=True)
    return sd[prefix + ".weight"]


def mlp(sd, prefix, layer_ids, x, hidden_act, out_act=None):
    """Sequential of Linear layers `<prefix>.<id>` with activation between."""
    for n, i in enumerate(layer_ids):
        W = wn_weight(sd, f"{prefix}.{i}")
        b = sd[f"{prefix}.{i}.bias"]
        x = torch.nn.functional.linear(x, W, b)
        if n + 1 < len(layer_ids):
            x = hidden_act(x)
    return out_act(x) if out_act is not None else x
