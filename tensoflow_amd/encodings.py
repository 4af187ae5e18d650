"""Differentiable encodings of the training direction (device-resident torch; the inference kernels evaluate the same
functions in registers: csrc/inner_light.hip, csrc/shape_shade.hip).

* `posenc`  -- get_embedder (utils/network_utils.py:38-50): [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(n-1) x), cos(2^(n-1) x)]
* `ide5`    -- generate_ide_fn(5) (utils/ref_utils.py:53-117): integrated directional encoding, 72 values
               [Re(38 terms... 36) | Im(36)], attenuated by exp(-l(l+1)/2 * kappa_inv)
* `linear_to_srgb` -- utils/raw_utils.py:4-17
"""
import math

import numpy as np
import torch


def posenc(x, n_freq):
    if x.is_cuda and x.dim() == 2 and not (torch.is_grad_enabled() and x.requires_grad) and x.dtype == torch.float32:
        from . import ops
        return ops.posenc_fwd(x, n_freq)                 # one launch (tf_posenc_fwd) instead of 2 n_freq + 1 + a concatenation
    out = [x]
    for k in range(n_freq):
        out += [torch.sin(x * float(2 ** k)), torch.cos(x * float(2 ** k))]
    return torch.cat(out, -1)


def _ide_tables():
    """-> mat [17, 36] (coefficient of z^k for column (l, m)), m_of_col [36], l_of_col [36]; l = 1, 2, 4, 8, 16, m = 0..l."""
    f = math.factorial
    cols = [(1 << d, m) for d in range(5) for m in range((1 << d) + 1)]
    mat = np.zeros((17, len(cols)))
    for c, (l, m) in enumerate(cols):
        for k in range(l - m + 1):
            a = 0.5 * (l + k + m - 1.0)
            gb = np.prod([a - j for j in range(l)]) / f(l)           # generalised binomial C(a, l)
            leg = (-1.0) ** m * 2.0 ** l * f(l) / f(k) / f(l - k - m) * gb
            mat[k, c] = math.sqrt((2.0 * l + 1.0) * f(l - m) / (4.0 * math.pi * f(l + m))) * leg
    return mat.astype(np.float32), np.array([m for _, m in cols]), np.array([l for l, _ in cols], np.float32)


_IDE = {}


def ide5(xyz, kappa_inv, wide=False):
    """xyz [..., 3] unit directions, kappa_inv [..., 1] -> [..., 72].
    wide: evaluate in fp64 (Horner on the fp32-ROUNDED coefficient table, which is part of the reference function) and return fp32.
    The degree-16 columns cancel catastrophically in fp32 -- the reference's own fp32 values are good to ~6e-4, their derivative wrt
    the direction to ~2e-3 (tools/gen_golden.py:gen_shading_direction measures both against an fp64 run) -- so where the training
    direction differentiates THROUGH the encoding (the 'direction' outer light under the roughness-warped fixed samplers) the wide
    form keeps this implementation's noise below the reference's instead of adding a second, independent copy of it."""
    if xyz.is_cuda and xyz.dim() == 2 and xyz.shape[1] == 3 and xyz.dtype == torch.float32 and _kappa_fits(xyz, kappa_inv):
        # round 4: one kernel forward, one backward (tf_ide5_fwd / tf_ide5_bwd: fp64 Horner on the fp32-rounded table, i.e. the `wide`
        # evaluation for every caller) instead of ~120 element-wise launches and autograd's ~250.  The kernel reads kappa[r] for every
        # row r: a broadcastable kappa_inv (0-dim, [1,1]) is expanded to one value per row first.
        if torch.is_tensor(kappa_inv) and kappa_inv.numel() != xyz.shape[0]:
            kappa_inv = kappa_inv.reshape(-1, 1).expand(xyz.shape[0], 1)
        return Ide5Fn.apply(xyz, kappa_inv if torch.is_tensor(kappa_inv) else None)
    if wide:
        return _ide5_wide(xyz, kappa_inv)
    dev = xyz.device
    if dev not in _IDE:
        mat, ms, ls = _ide_tables()
        _IDE[dev] = (torch.from_numpy(mat).to(dev), torch.from_numpy(ms).to(dev), torch.from_numpy(0.5 * ls * (ls + 1)).to(dev))
    mat, ms, sigma = _IDE[dev]
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    vmz = torch.cat([torch.ones_like(z)] + [z ** i for i in range(1, mat.shape[0])], -1)
    re, im = [torch.ones_like(x)], [torch.zeros_like(x)]                 # (x + iy)^m, m = 0..16
    for _ in range(16):
        re, im = re + [re[-1] * x - im[-1] * y], im + [re[-1] * y + im[-1] * x]
    re, im = torch.cat(re, -1)[..., ms], torch.cat(im, -1)[..., ms]
    if vmz.is_cuda and vmz.dim() == 2:     # [n,17] x [17,36]: a dense layer without bias on the HIP kernels (fwd + bwd), not a library GEMM
        from .autograd import LinearActFn
        from . import ops
        poly = LinearActFn.apply(vmz.contiguous(), mat.t().contiguous(), None, ops.ACT_NONE, 0.0)
    else:
        poly = vmz @ mat
    att = torch.exp(-sigma * kappa_inv)
    return torch.cat([re * poly * att, im * poly * att], -1)


def _kappa_fits(xyz, kappa_inv):
    """The fused kernels take kappa_inv as None (== 0) or as ONE fp32 value per row on xyz's device (after broadcasting)."""
    if not torch.is_tensor(kappa_inv):
        return kappa_inv == 0
    return (kappa_inv.dtype == torch.float32 and kappa_inv.device == xyz.device and kappa_inv.numel() in (1, xyz.shape[0])
            and (kappa_inv.dim() <= 1 or kappa_inv.shape[-1] == 1))


class Ide5Fn(torch.autograd.Function):
    """ide5 on the device: forward tf_ide5_fwd, backward tf_ide5_bwd (closed form wrt the direction and kappa_inv)."""

    @staticmethod
    def _coef(dev):
        key = (dev, "coef")
        if key not in _IDE:
            _IDE[key] = torch.from_numpy(_ide_tables()[0]).to(dev).contiguous()
        return _IDE[key]

    @staticmethod
    def forward(ctx, xyz, kappa_inv):
        from . import ops
        xyz = xyz.contiguous()
        if xyz.dim() != 2 or xyz.shape[1] != 3 or xyz.dtype != torch.float32:
            raise RuntimeError(f"Ide5Fn: xyz must be fp32 [n,3], got {xyz.dtype} {tuple(xyz.shape)}")
        if kappa_inv is not None and (kappa_inv.numel() != xyz.shape[0] or kappa_inv.dtype != torch.float32 or kappa_inv.device != xyz.device):
            raise RuntimeError(f"Ide5Fn: kappa_inv must hold one fp32 value per row on {xyz.device}, got {kappa_inv.dtype} "
                               f"{tuple(kappa_inv.shape)} on {kappa_inv.device} for {xyz.shape[0]} rows")
        kap = None if kappa_inv is None else kappa_inv.reshape(-1).contiguous()
        ctx.save_for_backward(xyz, kap if kap is not None else torch.empty(0, device=xyz.device))
        ctx.has_kap = kap is not None
        ctx.kshape = None if kappa_inv is None else kappa_inv.shape
        return ops.ide5_fwd(xyz, kap, Ide5Fn._coef(xyz.device))

    @staticmethod
    def backward(ctx, g):
        from . import ops
        xyz, kap = ctx.saved_tensors
        want_k = ctx.has_kap and ctx.needs_input_grad[1]
        g_xyz, g_k = ops.ide5_bwd(xyz, kap if ctx.has_kap else None, Ide5Fn._coef(xyz.device), g.contiguous(), want_kappa=want_k)
        return (g_xyz if ctx.needs_input_grad[0] else None), (g_k.reshape(ctx.kshape) if (want_k and g_k is not None) else None)


def _ide5_wide(xyz, kappa_inv):
    dev = xyz.device
    key = (dev, "wide")
    if key not in _IDE:
        mat, ms, ls = _ide_tables()
        _IDE[key] = (torch.from_numpy(mat.astype(np.float64)).to(dev), torch.from_numpy(ms).to(dev), torch.from_numpy(0.5 * ls * (ls + 1)).double().to(dev))
    mat, ms, sigma = _IDE[key]
    q = xyz.double()
    x, y, z = q[..., 0:1], q[..., 1:2], q[..., 2:3]
    poly = mat[16].expand(*z.shape[:-1], 36)
    for k in range(15, -1, -1):                 # Horner, all 36 columns at once (element-wise kernels: no library GEMM)
        poly = poly * z + mat[k]
    re, im = [torch.ones_like(x)], [torch.zeros_like(x)]
    for _ in range(16):
        re, im = re + [re[-1] * x - im[-1] * y], im + [re[-1] * y + im[-1] * x]
    re, im = torch.cat(re, -1)[..., ms], torch.cat(im, -1)[..., ms]
    kap = kappa_inv.double() if torch.is_tensor(kappa_inv) else torch.as_tensor(float(kappa_inv), dtype=torch.float64, device=dev)
    att = torch.exp(-sigma * kap)
    return torch.cat([re * poly * att, im * poly * att], -1).float()


def linear_to_srgb(lin):
    eps = torch.finfo(torch.float32).eps
    return torch.where(lin <= 0.0031308, 323 / 25 * lin, (211 * lin.clamp(min=eps) ** (5 / 12) - 11) / 200)
