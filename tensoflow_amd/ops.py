"""Thin torch-tensor wrappers over the C ABI (one function per entry point).

PyTorch is plumbing here: it owns device memory and the stream; every op below validates
its operands on the host (device, dtype, contiguity, shapes) BEFORE the kernel is launched,
then enqueues on the current HIP stream.  No op has a PyTorch fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L


PREC_F32, PREC_F16X3, PREC_F16, PREC_F16X2, PREC_BF16X3 = 0, 1, 2, 3, 4      # TfPrecision (PREC_F16: flow + inner-light decoders only, outside the 1e-4 bar; PREC_BF16X3: dense layers only)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (tensoflow_amd has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


_AABB_HOST = {}


def _aabb6(aabb):
    """The box as six host floats (a kernel argument by value).  A box that lives on the device is read back ONCE per tensor OBJECT and
    version: every call used to be a device sync -- 14 of them in a shape-stage training step.  The cache entry holds the tensor itself
    (its storage stays alive, so the allocator cannot hand the same address to another box) and is valid only for that very object at
    that version; a raw (data_ptr, version) key does not identify a tensor (advisor, round 3)."""
    if torch.is_tensor(aabb) and aabb.is_cuda:
        ent = _AABB_HOST.get(id(aabb))
        if ent is None or ent[0] is not aabb or ent[1] != aabb._version:
            if len(_AABB_HOST) > 64:
                _AABB_HOST.clear()
            ent = _AABB_HOST[id(aabb)] = (aabb, aabb._version, aabb.detach().float().reshape(-1).cpu().tolist())
        a = ent[2]
    else:
        a = torch.as_tensor(aabb, dtype=torch.float32).reshape(-1).tolist()
    return (C.c_float * 6)(*a)


def _f(t):
    return t.contiguous().float()


# ------------------------------------------------------------------------------ VM field
class VmPacked:
    """Channel-last packed mip pyramid of a VM field (built by tf_vm_pack_fwd)."""

    def __init__(self, planes, lines, n_levels, texel_f16=False):
        """texel_f16: BASELINE configs[4]'s "fp16 field" -- the pyramid the kernels read holds IEEE halves (tf_vm_pack_to_f16 of the
        fp32 pyramid: mips averaged in fp32, rounded once; half the bytes per texel), taps widened to fp32 on load.  Inference only,
        opt-in, not parity-grade."""
        self.lib = L.load()
        C_ = planes[0].shape[1]
        d = L.TfVmDesc()
        d.C, d.n_levels = C_, n_levels
        d.texel_f16 = 1 if texel_f16 else 0
        self.texel_f16 = bool(texel_f16)
        for i in range(3):
            assert planes[i].shape[0] == 1 and lines[i].shape[0] == 1 and lines[i].shape[3] == 1
            assert planes[i].shape[1] == C_ and lines[i].shape[1] == C_
            d.ph[i], d.pw[i], d.ll[i] = planes[i].shape[2], planes[i].shape[3], lines[i].shape[2]
        self.desc = d
        n = self.lib.tf_vm_packed_floats(C.byref(d))
        if n == 0:
            raise RuntimeError("tf_vm_packed_floats: invalid field geometry (C % 4, n_levels, divisibility)")
        self.n_floats = n
        self.C = C_
        self.n_levels = n_levels
        self.data = torch.empty(n, dtype=torch.float32, device=planes[0].device)
        self.data16 = torch.empty(n, dtype=torch.float16, device=planes[0].device) if texel_f16 else None
        self.repack(planes, lines)
        if texel_f16:
            self.data = None                    # only the half pyramid stays resident

    def ptr(self):
        """Device pointer of the pyramid the kernels gather from (fp32, or halves when desc.texel_f16)."""
        return _p(self.data16, torch.float16) if self.texel_f16 else _p(self.data)

    def repack(self, planes, lines):
        pl = [_f(p.detach()) for p in planes]
        ln = [_f(l.detach()) for l in lines]
        pa = (C.c_void_p * 3)(*[p.data_ptr() for p in pl])
        la = (C.c_void_p * 3)(*[l.data_ptr() for l in ln])
        data = self.data if self.data is not None else torch.empty(self.n_floats, dtype=torch.float32, device=self.data16.device)
        L.check(self.lib.tf_vm_pack_fwd(C.byref(self.desc), C.byref(pa), C.byref(la), _p(data), _stream()),
                "tf_vm_pack_fwd")
        if self.texel_f16:
            L.check(self.lib.tf_vm_pack_to_f16(C.byref(self.desc), _p(data), _p(self.data16, torch.float16), _stream()), "tf_vm_pack_to_f16")

    def unpack_grad(self, gpacked, planes, lines):
        gp = [torch.empty_like(p, memory_format=torch.contiguous_format) for p in planes]
        gl = [torch.empty_like(l, memory_format=torch.contiguous_format) for l in lines]
        pa = (C.c_void_p * 3)(*[p.data_ptr() for p in gp])
        la = (C.c_void_p * 3)(*[l.data_ptr() for l in gl])
        L.check(self.lib.tf_vm_pack_bwd(C.byref(self.desc), _p(gpacked), C.byref(pa), C.byref(la), _stream()),
                "tf_vm_pack_bwd")
        return gp, gl


def vm_gather(packed: VmPacked, xyz, level, aabb):
    xyz = _f(xyz)
    n = xyz.shape[0]
    feat = torch.empty(n, 3 * packed.C, dtype=torch.float32, device=xyz.device)
    lv = None if level is None else _f(level.reshape(-1))
    L.check(packed.lib.tf_vm_gather_fwd(C.byref(packed.desc), packed.ptr(), _p(xyz), _p(lv), C.byref(_aabb6(aabb)), n,
                                        _p(feat), _stream()), "tf_vm_gather_fwd")
    return feat


def vm_gather_bwd(packed: VmPacked, xyz, level, aabb, gfeat):
    xyz = _f(xyz)
    gfeat = _f(gfeat)
    if packed.texel_f16:
        raise RuntimeError("vm_gather_bwd: a half pyramid (texel_f16) is an inference-only format")
    gpacked = torch.zeros_like(packed.data)
    lv = None if level is None else _f(level.reshape(-1))
    L.check(packed.lib.tf_vm_gather_bwd(C.byref(packed.desc), _p(packed.data), _p(xyz), _p(lv), C.byref(_aabb6(aabb)),
                                        xyz.shape[0], _p(gfeat), _p(gpacked), _stream()), "tf_vm_gather_bwd")
    return gpacked


# ------------------------------------------------------------------------------ SDF decoder
def _sdf_mlp(w1, b1, w2, b2):
    m = L.TfSdfMlp()
    keep = [_f(w1), _f(b1), _f(w2), _f(b2)]
    m.w1, m.b1, m.w2, m.b2 = [t.data_ptr() for t in keep]
    m.hidden, m.app_dim = w1.shape[0], w2.shape[0] - 1
    return m, keep


def set_launch_budget(bvh_blocks_per_cu=0, flow_waves_per_block=0, inner_teams=0):
    """tf_set_launch_budget: how much of a CU the traversal / flow / inner-light kernels launched next (by this thread) take; 0 = default."""
    L.check(L.load().tf_set_launch_budget(int(bvh_blocks_per_cu), int(flow_waves_per_block), int(inner_teams)), "tf_set_launch_budget")


def probe_mfma_f16_tflops(iters=200000, device=None, relu_like=False):
    """tf_probe_mfma_f16: executed TFLOP/s the matrix cores of this device sustain on a dense v_mfma_f32_32x32x16_f16 stream with random
    operands (relu_like: the activation operands half zero, as behind a ReLU) (synchronous; ~0.2 s at the default).  A measurement aid
    of bench.py's roofline figures, never on a product path."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = 256 * torch.cuda.get_device_properties(dev).multi_processor_count
    scratch = torch.empty(n, dtype=torch.float32, device=dev)
    out = C.c_double(0.0)
    L.check(L.load().tf_probe_mfma_f16(int(iters), int(bool(relu_like)), _p(scratch), n, C.byref(out), _stream()), "tf_probe_mfma_f16")
    return float(out.value)


_ws_cache = {}


def _workspace(key, n_floats, device):
    t = _ws_cache.get((key, str(device)))
    if t is None or t.numel() < n_floats:
        t = torch.empty(int(n_floats), dtype=torch.float32, device=device)
        _ws_cache[(key, str(device))] = t
    return t


def sdf_embed_freqs(packed: VmPacked, w1):
    """sdf_multires of a TensoSDF decoder from the width of its first layer: 3C + 3 + 6 m inputs (fields.py:66-81: get_embedder(m, 3)
    appended to the VM features).  0 = the shipped configs, served by the fused kernels; m > 0 by the composition below."""
    extra = w1.shape[1] - 3 * packed.desc.C - 3
    if extra < 0 or extra % 6:
        raise RuntimeError(f"TensoSDF decoder: {w1.shape[1]} inputs for {packed.desc.C} components per plane")
    return extra // 6


def _sdf_decode_composed(packed, w1, b1, w2, b2, xyz, level, aabb, rows_out=None):
    """TensoSDF.forward with sdf_multires = m > 0 (fields.py:293-299): [VM features | embed(xyz)] -> Linear -> Softplus(100) -> Linear, as a
    composition of the gather kernel, the positional-encoding kernel and the dense-layer kernels (exact fp32).  The reference embeds
    the CONTRACTED coordinates when m == 3 and the raw ones otherwise (:294).  rows_out: use only these rows of the second layer."""
    from .encodings import posenc
    m = sdf_embed_freqs(packed, w1)
    feat = vm_gather(packed, xyz, level, aabb)
    a = torch.as_tensor(aabb, dtype=torch.float32, device=xyz.device)
    e = posenc(((xyz - a[0]) / (a[1] - a[0])) if m == 3 else xyz, m)
    h = linear_fwd(torch.cat([feat, e], -1).contiguous(), w1, b1, ACT_SOFTPLUS, 100.0)
    if rows_out is not None:
        w2, b2 = w2[rows_out].contiguous(), b2[rows_out].contiguous()
    return linear_fwd(h, w2, b2)


def sdf_forward(packed: VmPacked, w1, b1, w2, b2, xyz, level, aabb, want_feat=True, precision=1):
    """TensoSDF.forward -> (sdf [n], feat [n,A] or None).  precision: PREC_F16X3 (default) or PREC_F32."""
    lib = packed.lib
    xyz = _f(xyz)
    if sdf_embed_freqs(packed, w1) > 0:
        out = _sdf_decode_composed(packed, w1, b1, w2, b2, xyz, level, aabb, rows_out=None if want_feat else slice(0, 1))
        return out[:, 0].contiguous(), (out[:, 1:].contiguous() if want_feat else None)
    n = xyz.shape[0]
    mlp, keep = _sdf_mlp(w1, b1, w2, b2)
    sdf = torch.empty(n, dtype=torch.float32, device=xyz.device)
    feat = torch.empty(n, w2.shape[0] - 1, dtype=torch.float32, device=xyz.device) if want_feat else None
    ws = _workspace("sdf", lib.tf_sdf_workspace_floats(), xyz.device)
    lv = None if level is None else _f(level.reshape(-1))
    L.check(lib.tf_sdf_forward(C.byref(packed.desc), packed.ptr(), C.byref(mlp), _p(xyz), _p(lv), C.byref(_aabb6(aabb)),
                               n, _p(sdf), _p(feat), int(precision), _p(ws), ws.numel(), _stream()), "tf_sdf_forward")
    return sdf, feat


def sdf_alpha(packed: VmPacked, w1, b1, w2, b2, pts, level, dists, dirs, aabb, units, inv_s, cos_anneal,
              want_feat=True, want_hess=True, precision=None, want_taps=False):
    """ShapeRenderer.compute_sdf_alpha -> alpha, grad, feat, sdf, normal_hessian.
    precision: PREC_F16X3 (default) / PREC_F32.  (The hessian term's second difference divides rounding noise by eps^2;
    against the reference it measures 1.1e-3 with either decoder -- the reference's own fp32 rounding dominates.)"""
    if precision is None:
        precision = PREC_F16X3
    lib = packed.lib
    pts, dists, dirs = _f(pts), _f(dists.reshape(-1)), _f(dirs)
    n = pts.shape[0]
    dev = pts.device
    if sdf_embed_freqs(packed, w1) > 0:
        # sdf_multires > 0: the seven evaluations through the composed decoder, the rest of compute_sdf_alpha element-wise
        # (shapeRenderer.py:995-1025; fields.py:227-260)
        u = torch.as_tensor([float(v) for v in units], dtype=torch.float32, device=dev)
        offs = torch.zeros(7, 3, device=dev)
        for ax in range(3):
            offs[1 + 2 * ax, ax], offs[2 + 2 * ax, ax] = u[ax], -u[ax]
        lv = None if level is None else _f(level.reshape(-1))
        c = _sdf_decode_composed(packed, w1, b1, w2, b2, pts, lv, aabb, rows_out=None if want_feat else slice(0, 1))
        sdf, feat = c[:, 0].contiguous(), (c[:, 1:].contiguous() if want_feat else None)
        P = (pts[None] + offs[1:, None]).reshape(-1, 3).contiguous()
        taps = _sdf_decode_composed(packed, w1, b1, w2, b2, P, None if lv is None else lv.repeat(6), aabb, rows_out=slice(0, 1)).view(6, n).t().contiguous()
        grad = torch.stack([(taps[:, 2 * ax] - taps[:, 2 * ax + 1]) / (2 * u[ax]) for ax in range(3)], -1)
        nh = None
        if want_hess:
            hess = torch.stack([(taps[:, 2 * ax] + taps[:, 2 * ax + 1] - 2 * sdf) / (u[ax] ** 2) for ax in range(3)], -1)
            nh = (grad * hess).sum(-1) / ((grad ** 2).sum(-1) + 1e-5)
        true_cos = (dirs * grad).sum(-1)
        iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + torch.relu(-true_cos) * cos_anneal)
        pc, nc = torch.sigmoid((sdf - iter_cos * dists * 0.5) * inv_s), torch.sigmoid((sdf + iter_cos * dists * 0.5) * inv_s)
        alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
        return (alpha, grad, feat, sdf, nh, taps) if want_taps else (alpha, grad, feat, sdf, nh)
    mlp, keep = _sdf_mlp(w1, b1, w2, b2)
    alpha = torch.empty(n, dtype=torch.float32, device=dev)
    grad = torch.empty(n, 3, dtype=torch.float32, device=dev)
    sdf = torch.empty(n, dtype=torch.float32, device=dev)
    feat = torch.empty(n, w2.shape[0] - 1, dtype=torch.float32, device=dev) if want_feat else None
    nh = torch.empty(n, dtype=torch.float32, device=dev) if want_hess else None
    taps = torch.empty(n, 6, dtype=torch.float32, device=dev) if want_taps else None      # the six FD sdf values, kept for sdf_alpha_bwd
    ws = _workspace("sdf", lib.tf_sdf_workspace_floats(), dev)
    lv = None if level is None else _f(level.reshape(-1))
    un = (C.c_float * 3)(*[float(u) for u in units])
    L.check(lib.tf_sdf_alpha_fwd(C.byref(packed.desc), packed.ptr(), C.byref(mlp), _p(pts), _p(lv), _p(dists), _p(dirs),
                                 C.byref(_aabb6(aabb)), C.byref(un), float(inv_s), float(cos_anneal), n, _p(alpha), _p(grad),
                                 _p(feat), _p(sdf), _p(nh), _p(taps), int(precision), _p(ws), ws.numel(), _stream()), "tf_sdf_alpha_fwd")
    if want_taps:
        return alpha, grad, feat, sdf, nh, taps
    return alpha, grad, feat, sdf, nh


def sdf_alpha_bwd(packed: VmPacked, w1, b1, w2, b2, pts, level, dists, dirs, aabb, units, inv_s, cos_anneal, sdf, taps,
                  g_alpha=None, g_grad=None, g_feat=None, g_sdf=None, g_nh=None, precision=None):
    """tf_sdf_alpha_bwd: backward of compute_sdf_alpha in one entry point -> (gpacked [pyramid], g_w1, g_b1, g_w2, g_b2, g_inv_s [1])."""
    if precision is None:
        precision = PREC_F16X3
    if packed.texel_f16:
        raise RuntimeError("sdf_alpha_bwd: a half pyramid (texel_f16) is an inference-only format")
    lib = packed.lib
    pts, dists, dirs = _f(pts), _f(dists.reshape(-1)), _f(dirs)
    n = pts.shape[0]
    dev = pts.device
    mlp, keep = _sdf_mlp(w1, b1, w2, b2)
    gpacked = torch.zeros_like(packed.data)
    g_w1, g_b1, g_w2, g_b2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)
    g_inv = torch.zeros(1, dtype=torch.float32, device=dev)
    ws = _workspace("sdf_bwd", lib.tf_sdf_alpha_bwd_workspace_floats(n), dev)
    lv = None if level is None else _f(level.reshape(-1))
    un = (C.c_float * 3)(*[float(u) for u in units])
    g = lambda t, shape: None if t is None else _f(t).reshape(shape)
    ga, gg, gf, gs, gn = g(g_alpha, (n,)), g(g_grad, (n, 3)), g(g_feat, (n, w2.shape[0] - 1)), g(g_sdf, (n,)), g(g_nh, (n,))
    L.check(lib.tf_sdf_alpha_bwd(C.byref(packed.desc), _p(packed.data), C.byref(mlp), _p(pts), _p(lv), _p(dists), _p(dirs),
                                 C.byref(_aabb6(aabb)), C.byref(un), float(inv_s), float(cos_anneal), n, _p(_f(sdf)), _p(_f(taps)),
                                 _p(ga), _p(gg), _p(gf), _p(gs), _p(gn), _p(gpacked), _p(g_w1), _p(g_b1), _p(g_w2), _p(g_b2), _p(g_inv),
                                 int(precision), _p(ws), ws.numel(), _stream()), "tf_sdf_alpha_bwd")
    return gpacked, g_w1, g_b1, g_w2, g_b2, g_inv


# ------------------------------------------------------------------------------ sample_ray (shapeRenderer.py:871-932)
_LIN_CACHE = {}


def _lin(key, fn, device):
    t = _LIN_CACHE.get((key, str(device)))
    if t is None:
        t = _LIN_CACHE[(key, str(device))] = fn().to(device).contiguous()
    return t


def sample_ray_init(o, d, near, far, radiis, rays_cos, aabb, n_samples, base_radii, t_rand=None):
    """tf_sample_ray_init -> z [rn,S], pts [rn*S,3], level [rn*S]."""
    lib = L.load()
    o, d = _f(o), _f(d)
    rn, dev = o.shape[0], o.device
    z = torch.empty(rn, n_samples, dtype=torch.float32, device=dev)
    pts = torch.empty(rn * n_samples, 3, dtype=torch.float32, device=dev)
    lv = torch.empty(rn * n_samples, dtype=torch.float32, device=dev)
    lin = _lin(("lin01", n_samples), lambda: torch.linspace(0.0, 1.0, n_samples), dev)
    L.check(lib.tf_sample_ray_init(_p(o), _p(d), _p(_f(near.reshape(-1))), _p(_f(far.reshape(-1))), _p(_f(radiis.reshape(-1))),
                                   _p(_f(rays_cos.reshape(-1))), C.byref(_aabb6(aabb)), _p(lin),
                                   _p(_f(t_rand.reshape(-1))) if t_rand is not None else None, rn, int(n_samples), float(base_radii),
                                   _p(z), _p(pts), _p(lv), _stream()), "tf_sample_ray_init")
    return z, pts, lv


def sample_ray_intervals(o, d, t, aabb):
    """tf_sample_ray_intervals: t [rn,S] -> t0, t1 [rn*S], inner [rn*S] uint8 (the interval's midpoint lies inside the aabb)."""
    o, d, t = _f(o), _f(d), _f(t)
    rn, S = t.shape
    t0 = torch.empty(rn * S, dtype=torch.float32, device=t.device)
    t1 = torch.empty_like(t0)
    inner = torch.empty(rn * S, dtype=torch.uint8, device=t.device)
    L.check(L.load().tf_sample_ray_intervals(_p(o), _p(d), _p(t), rn, S, C.byref(_aabb6(aabb)), _p(t0), _p(t1), _p(inner, torch.uint8),
                                             _stream()), "tf_sample_ray_intervals")
    return t0, t1, inner


def sample_points(o, d, radiis, rays_cos, ray_indices, t0, t1, base_radii):
    """tf_sample_points -> mid [n], dists [n], viewdir [n,3], points [n,3], level [n,1] of the packed samples."""
    o, d, t0, t1 = _f(o), _f(d), _f(t0), _f(t1)
    n, dev = t0.shape[0], t0.device
    mid, dists, level = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3))
    viewdir, points = torch.empty(n, 3, dtype=torch.float32, device=dev), torch.empty(n, 3, dtype=torch.float32, device=dev)
    L.check(L.load().tf_sample_points(_p(o), _p(d), _p(_f(radiis.reshape(-1))), _p(_f(rays_cos.reshape(-1))),
                                      _p(ray_indices.contiguous(), torch.int64), _p(t0), _p(t1), n, float(base_radii), _p(mid), _p(dists),
                                      _p(viewdir), _p(points), _p(level), _stream()), "tf_sample_points")
    return mid, dists, viewdir, points, level.view(n, 1)


def sample_ray_upsample(o, d, radiis, rays_cos, z, sdf, n_imp, inv_s, base_radii, want_pts=True):
    """tf_sample_ray_upsample -> new_t [rn,n_imp] (, pts [rn*n_imp,3], level [rn*n_imp])."""
    lib = L.load()
    rn, S = z.shape
    dev = z.device
    new_t = torch.empty(rn, n_imp, dtype=torch.float32, device=dev)
    npts = torch.empty(rn * n_imp, 3, dtype=torch.float32, device=dev) if want_pts else None
    nlv = torch.empty(rn * n_imp, dtype=torch.float32, device=dev) if want_pts else None
    u = _lin(("u", n_imp), lambda: torch.linspace(0.5 / n_imp, 1.0 - 0.5 / n_imp, steps=n_imp), dev)
    L.check(lib.tf_sample_ray_upsample(_p(_f(o)), _p(_f(d)), _p(_f(radiis.reshape(-1))), _p(_f(rays_cos.reshape(-1))), _p(_f(z)), _p(_f(sdf)),
                                       rn, S, int(n_imp), float(inv_s), _p(u), float(base_radii), _p(new_t), _p(npts), _p(nlv), _stream()),
            "tf_sample_ray_upsample")
    return new_t, npts, nlv


def sample_ray_merge(z, sdf, new_t, new_sdf=None):
    """tf_sample_ray_merge -> z_out [rn, S + n_imp] (, sdf_out)."""
    lib = L.load()
    rn, S = z.shape
    n_imp = new_t.shape[1]
    z_out = torch.empty(rn, S + n_imp, dtype=torch.float32, device=z.device)
    sdf_out = torch.empty_like(z_out) if new_sdf is not None else None
    L.check(lib.tf_sample_ray_merge(_p(_f(z)), _p(_f(sdf)) if new_sdf is not None else None, _p(_f(new_t)),
                                    _p(_f(new_sdf)) if new_sdf is not None else None, rn, S, n_imp, _p(z_out), _p(sdf_out), _stream()),
            "tf_sample_ray_merge")
    return z_out, sdf_out


# ------------------------------------------------------------------------------ encodings of the training direction
def ide5_fwd(xyz, kappa_inv, coef):
    """tf_ide5_fwd: xyz [n,3], kappa_inv [n] or None, coef [17,36] -> [n,72]."""
    n = xyz.shape[0]
    out = torch.empty(n, 72, dtype=torch.float32, device=xyz.device)
    L.check(L.load().tf_ide5_fwd(_p(_f(xyz)), _p(_f(kappa_inv.reshape(-1))) if kappa_inv is not None else None, _p(coef), n, _p(out), _stream()),
            "tf_ide5_fwd")
    return out


def ide5_bwd(xyz, kappa_inv, coef, g_out, want_kappa=True):
    """tf_ide5_bwd -> (g_xyz [n,3], g_kappa [n] or None)."""
    n = xyz.shape[0]
    g_xyz = torch.empty(n, 3, dtype=torch.float32, device=xyz.device)
    g_k = torch.empty(n, dtype=torch.float32, device=xyz.device) if (want_kappa and kappa_inv is not None) else None
    L.check(L.load().tf_ide5_bwd(_p(_f(xyz)), _p(_f(kappa_inv.reshape(-1))) if kappa_inv is not None else None, _p(coef), _p(_f(g_out)), n,
                                 _p(g_xyz), _p(g_k), _stream()), "tf_ide5_bwd")
    return g_xyz, g_k


def posenc_fwd(x, n_freq):
    """tf_posenc_fwd: x [n,d] -> [n, d (1 + 2 n_freq)]."""
    x = _f(x)
    n, d = x.shape
    out = torch.empty(n, d * (1 + 2 * n_freq), dtype=torch.float32, device=x.device)
    L.check(L.load().tf_posenc_fwd(_p(x), n, d, int(n_freq), _p(out), _stream()), "tf_posenc_fwd")
    return out


def linear_to_srgb(lin, clamp01=False, g_out=None):
    """tf_linear_to_srgb_fwd (g_out None) / _bwd: element-wise on any shape."""
    lin = _f(lin)
    out = torch.empty_like(lin)
    if g_out is not None and g_out.shape != lin.shape:
        raise RuntimeError(f"linear_to_srgb: g_out {tuple(g_out.shape)} does not match lin {tuple(lin.shape)}")
    if g_out is None:
        L.check(L.load().tf_linear_to_srgb_fwd(_p(lin), lin.numel(), int(bool(clamp01)), _p(out), _stream()), "tf_linear_to_srgb_fwd")
    else:
        L.check(L.load().tf_linear_to_srgb_bwd(_p(lin), _p(_f(g_out)), lin.numel(), int(bool(clamp01)), _p(out), _stream()),
                "tf_linear_to_srgb_bwd")
    return out


# ------------------------------------------------------------------------------ split-sum shading algebra (training direction)
def shape_glue_pre(normals, view, mat, grads=None, mip_levels=None):
    """tf_shape_glue_pre_fwd: -> (normals_u [n,3], view_u [n,3], nov [n,1], reflective [n,3], roughness [n,1], mip [n] or None); with
    grads = (g_normals_u, g_nov, g_reflective, g_roughness, g_mip) (entries may be None) tf_shape_glue_pre_bwd: -> (g_normals [n,3],
    g_mat [n,5]).  mip_levels = (min_roughness, max_roughness, n_levels): also the specular-stack coordinate of EnvLight.get_mip."""
    normals, view = _f(normals), _f(view)
    mat = _f(mat) if mat is not None else None          # (the adjoint reads it for the mip coordinate only)
    n = normals.shape[0]
    if normals.shape != (n, 3) or view.shape != (n, 3) or (mat is not None and mat.shape != (n, 5)) or (mat is None and grads is None):
        raise RuntimeError(f"shape_glue_pre: normals {tuple(normals.shape)}, view {tuple(view.shape)}, mat {None if mat is None else tuple(mat.shape)}")
    dev = normals.device
    mn, mx, nl = (float(mip_levels[0]), float(mip_levels[1]), int(mip_levels[2])) if mip_levels is not None else (0.0, 0.5, 0)
    if grads is None:
        nu, vu, refl = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(3))
        nov, rough = (torch.empty(n, 1, dtype=torch.float32, device=dev) for _ in range(2))
        mip = torch.empty(n, dtype=torch.float32, device=dev) if mip_levels is not None else None
        L.check(L.load().tf_shape_glue_pre_fwd(_p(normals), _p(view), _p(mat), n, _p(nu), _p(vu), _p(nov), _p(refl), _p(rough), _p(mip), mn, mx, nl,
                                               _stream()), "tf_shape_glue_pre_fwd")
        return nu, vu, nov, refl, rough, mip
    g = [None if t is None else _f(t) for t in grads]
    if g[4] is not None and (mat is None or mip_levels is None):
        raise RuntimeError("shape_glue_pre: the mip coordinate's gradient needs mat and mip_levels")
    gn, gm = torch.empty(n, 3, dtype=torch.float32, device=dev), torch.empty(n, 5, dtype=torch.float32, device=dev)
    L.check(L.load().tf_shape_glue_pre_bwd(_p(normals), _p(view), _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]), _p(g[4]), _p(mat), mn, mx, nl, n, _p(gn),
                                           _p(gm), _stream()), "tf_shape_glue_pre_bwd")
    return gn, gm


def normalize3(x, acc=None, blend_c=None, want_err=False, grads=None):
    """tf_normalize3_fwd: F.normalize of [n,3] rows (of x acc + (1 - acc) blend_c with acc [n]) -> (y [n,3], err [n] = (|x| - 1)^2 or None);
    with grads = (g_y or None, g_err or None) tf_normalize3_bwd: -> (g_x [n,3], g_acc [n] or None)."""
    x = _f(x)
    n = x.shape[0]
    if x.shape != (n, 3) or (acc is not None and (acc.numel() != n or blend_c is None)):
        raise RuntimeError(f"normalize3: x {tuple(x.shape)}, acc {None if acc is None else tuple(acc.shape)}")
    acc = _f(acc) if acc is not None else None
    c = (C.c_float * 3)(*[float(v) for v in blend_c]) if blend_c is not None else None
    dev = x.device
    if grads is None:
        y = torch.empty(n, 3, dtype=torch.float32, device=dev)
        err = torch.empty(n, dtype=torch.float32, device=dev) if want_err else None
        L.check(L.load().tf_normalize3_fwd(_p(x), _p(acc), c, n, _p(y), _p(err), _stream()), "tf_normalize3_fwd")
        return y, err
    gy, ge = (None if t is None else _f(t) for t in grads)
    gx = torch.empty(n, 3, dtype=torch.float32, device=dev)
    ga = torch.empty(n, dtype=torch.float32, device=dev) if acc is not None else None
    L.check(L.load().tf_normalize3_bwd(_p(x), _p(acc), c, _p(gy), _p(ge), n, _p(gx), _p(ga), _stream()), "tf_normalize3_bwd")
    return gx, ga


def shape_glue_post(mat, nov, diffuse_light, direct_light, indirect_light, occ_raw, fg_lut, grads=None):
    """tf_shape_glue_post_fwd: -> (color [n,3], occ_prob [n,1]); with grads = (g_color, g_occ_prob or None) tf_shape_glue_post_bwd:
    -> (g_mat [n,5], g_nov [n,1], g_diffuse_light, g_direct_light, g_indirect_light [n,3], g_occ_raw [n,1]).  fg_lut [.., H, W, 2]."""
    mat, nov, dl, dr, il, oc, lut = _f(mat), _f(nov), _f(diffuse_light), _f(direct_light), _f(indirect_light), _f(occ_raw), _f(fg_lut)
    n = mat.shape[0]
    H, W = int(lut.shape[-3]), int(lut.shape[-2])
    if mat.shape != (n, 5) or nov.numel() != n or oc.numel() != n or any(t.shape != (n, 3) for t in (dl, dr, il)) or lut.shape[-1] != 2:
        raise RuntimeError("shape_glue_post: operand shapes")
    dev = mat.device
    if grads is None:
        color, occ_prob = torch.empty(n, 3, dtype=torch.float32, device=dev), torch.empty(n, 1, dtype=torch.float32, device=dev)
        L.check(L.load().tf_shape_glue_post_fwd(_p(mat), _p(nov), _p(dl), _p(dr), _p(il), _p(oc), _p(lut), H, W, n, _p(color), _p(occ_prob),
                                                _stream()), "tf_shape_glue_post_fwd")
        return color, occ_prob
    g_color = _f(grads[0])
    g_occ = None if grads[1] is None else _f(grads[1])
    gm = torch.empty(n, 5, dtype=torch.float32, device=dev)
    gnov, gocc = (torch.empty(n, 1, dtype=torch.float32, device=dev) for _ in range(2))
    gdl, gdr, gil = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(3))
    L.check(L.load().tf_shape_glue_post_bwd(_p(mat), _p(nov), _p(dl), _p(dr), _p(il), _p(oc), _p(lut), H, W, _p(g_color), _p(g_occ), n, _p(gm),
                                            _p(gnov), _p(gdl), _p(gdr), _p(gil), _p(gocc), _stream()), "tf_shape_glue_post_bwd")
    return gm, gnov, gdl, gdr, gil, gocc


# ------------------------------------------------------------------------------ compositing
def composite(alpha, ray_indices, values, n_rays):
    lib = L.load()
    alpha = _f(alpha)
    n = alpha.shape[0]
    k = 0 if values is None else values.shape[1]
    values = None if values is None else _f(values)
    dev = alpha.device
    w = torch.empty(n, dtype=torch.float32, device=dev)
    acc = torch.empty(n_rays, dtype=torch.float32, device=dev)
    out = torch.empty(n_rays, k, dtype=torch.float32, device=dev)
    L.check(lib.tf_composite_fwd(_p(alpha), _p(ray_indices, torch.int64), _p(values), n, n_rays, k, _p(w), _p(acc), _p(out),
                                 _stream()), "tf_composite_fwd")
    return w, acc, out


def composite_bwd(alpha, ray_indices, values, weights, g_acc, g_out, n_rays):
    lib = L.load()
    n = alpha.shape[0]
    k = 0 if values is None else values.shape[1]
    g_alpha = torch.empty_like(alpha)
    g_values = None if values is None else torch.empty_like(values)
    L.check(lib.tf_composite_bwd(_p(alpha), _p(ray_indices, torch.int64), _p(values), _p(weights),
                                 _p(None if g_acc is None else _f(g_acc)), _p(None if g_out is None else _f(g_out)), n, n_rays,
                                 k, _p(g_alpha), _p(g_values), _stream()), "tf_composite_bwd")
    return g_alpha, g_values


# ------------------------------------------------------------------------------ flow
def _coupling_nets(weights):
    """weights: list of 2 lists of 4 (W, b) pairs."""
    nets = (L.TfCouplingNet * 2)()
    keep = []
    for b in range(2):
        for l in range(4):
            W, bias = _f(weights[b][l][0]), _f(weights[b][l][1])
            keep += [W, bias]
            nets[b].w[l] = W.data_ptr()
            nets[b].b[l] = bias.data_ptr()
    expect = [(64, 44), (64, 64), (64, 64), (21, 64)]
    for b in range(2):
        for l in range(4):
            if tuple(weights[b][l][0].shape) != expect[l]:
                raise RuntimeError(f"coupling net {b} layer {l}: weight shape {tuple(weights[b][l][0].shape)} != {expect[l]}")
    return nets, keep


def flow_sample(weights, cond, latent, jitter=None, want_bins=False, precision=1, cache=None):
    """-> angles [pn,sn,2], logj [pn,sn,1] (, bins [pn,sn,2] int32).  cache: optional PackCache owned by this flow."""
    lib = L.load()
    cond, latent = _f(cond), _f(latent)
    pn, sn = cond.shape[0], latent.shape[0]
    if cond.shape[1] != 37:
        raise RuntimeError("cond must be [pn,37]")
    dev = cond.device
    nets, keep = _coupling_nets(weights)
    ang = torch.empty(pn, sn, 2, dtype=torch.float32, device=dev)
    lj = torch.empty(pn, sn, 1, dtype=torch.float32, device=dev)
    bins = torch.empty(pn, sn, 2, dtype=torch.int32, device=dev) if want_bins else None
    if cache is None:
        ws, flag = _workspace("flow", lib.tf_flow_workspace_floats(pn), dev), 0
    else:
        ws = cache.workspace(lib.tf_flow_workspace_floats(pn), dev)
        flag = cache.flag(keep, precision)
    jit = None if jitter is None else _f(jitter.reshape(pn, sn))
    L.check(lib.tf_flow_sample_fwd(C.byref(nets), _p(cond), _p(latent), _p(jit), pn, sn, _p(ang), _p(lj), _p(bins, torch.int32),
                                   int(precision) | flag, _p(ws), ws.numel(), _stream()), "tf_flow_sample_fwd")
    return (ang, lj, bins) if want_bins else (ang, lj)


ACT_NONE, ACT_RELU, ACT_SOFTPLUS, ACT_SIGMOID, ACT_EXP_CLAMP = 0, 1, 2, 3, 4      # TfActivation


LINEAR_PRECISION = PREC_BF16X3   # operand arithmetic of the training direction's dense layers: fp32-grade with fp32's range (a bf16 triple split on the aligned shapes -- six bf16 matrix steps per 16-deep product -- the exact-fp32 matrix instruction elsewhere).  PREC_F32 = the exact instruction everywhere (round 6: named by the caller, no environment switch).  PREC_F16X3 exists in the
                                 # kernel and is NOT the default for two measured reasons: gradients of mean-reduced losses (1e-7 .. 1e-5 per
                                 # element) fall below the f16 range and flush to zero unscaled (test_mcshading_eval_follows_parameter_updates
                                 # caught it), and the tall-skinny products are held by their tile traffic, not by the matrix rate (material
                                 # training step 17.9 vs 18.1 ms)


def linear_fwd(x, w, b, act=ACT_NONE, act_param=0.0, n_dev=None, precision=None):
    """Y = act(x w^T + b) on the matrix cores (tf_linear_fwd): x [n,K], w [N,K], b [N] or None -> [n,N].
    precision: PREC_BF16X3 (fp32-grade, see LINEAR_PRECISION; the module default), PREC_F32 (the exact-fp32 instruction) or PREC_F16X3 (operands within the f16 range only).
    n_dev: device int64 scalar -- only the first min(n, n_dev) rows are computed (the rest of Y stays uninitialised)."""
    lib = L.load()
    x, w = _f(x), _f(w)
    n, K = x.shape
    N = w.shape[0]
    assert w.shape == (N, K)
    y = torch.empty(n, N, device=x.device)
    L.check(lib.tf_linear_fwd(_p(x), _p(w), _p(_f(b)) if b is not None else None, n, K, N, int(act), float(act_param),
                              int(LINEAR_PRECISION if precision is None else precision), _p(y),
                              _p(n_dev, torch.int64) if n_dev is not None else None, _stream()), "tf_linear_fwd")
    return y


def linear_bwd(x, w, y, gy, act=ACT_NONE, act_param=0.0, need_gx=True, need_gw=True, need_gb=True, n_dev=None, precision=None):
    """-> (gx [n,K] | None, gw [N,K] | None, gb [N] | None) of Y = act(x w^T + b) given the forward output y and gy (tf_linear_bwd)."""
    lib = L.load()
    x, w, y, gy = _f(x), _f(w), _f(y), _f(gy)
    n, K = x.shape
    N = w.shape[0]
    gz = torch.empty(n, N, device=x.device)
    gx = torch.empty(n, K, device=x.device) if need_gx else None
    gw = torch.empty(N, K, device=x.device) if need_gw else None
    gb = torch.empty(N, device=x.device) if need_gb else None
    L.check(lib.tf_linear_bwd(_p(x), _p(w), _p(y), _p(gy), n, K, N, int(act), float(act_param),
                              int(LINEAR_PRECISION if precision is None else precision), _p(gz), _p(gx) if need_gx else None,
                              _p(gw) if need_gw else None, _p(gb) if need_gb else None,
                              _p(n_dev, torch.int64) if n_dev is not None else None, _stream()), "tf_linear_bwd")
    return gx, gw, gb


def linear_bwd_fused(x, w, y, gy, act=ACT_NONE, act_param=0.0, gy_is_gz=False, x_act=ACT_NONE, x_act_param=0.0, need_gx=True,
                     need_gw=True, need_gb=True, need_gbx=False, n_dev=None, precision=None, zeroed=None):
    """One layer of a backward chain (tf_linear_bwd_fused) -> (gx, gw, gb, gbx).  With x_act the returned gx is the gradient wrt the
    PRE-activation of the layer below (whose output x is) and gbx that layer's bias gradient; pass it on with gy_is_gz=True.
    zeroed: a callable n_floats -> zero-filled float tensor carved out of ONE buffer the caller filled once for the whole chain
    (TF_BWD_GRADS_ZEROED: no fill launches in here)."""
    lib = L.load()
    x, w, gy = _f(x), _f(w), _f(gy)
    n, K = x.shape
    N = w.shape[0]
    y = None if gy_is_gz else _f(y)
    gz = None if gy_is_gz else torch.empty(n, N, device=x.device)
    gx = torch.empty(n, K, device=x.device) if need_gx else None
    new = (lambda k: zeroed(k)) if zeroed is not None else (lambda k: torch.empty(k, device=x.device))
    gw = new(N * K).view(N, K) if need_gw else None
    gb = new(N) if (need_gb and not gy_is_gz) else None
    gbx = new(K) if (need_gbx and need_gx) else None
    L.check(lib.tf_linear_bwd_fused(_p(x), _p(w), _p(y) if y is not None else None, _p(gy), n, K, N, int(act), float(act_param),
                                    (1 if gy_is_gz else 0) | (2 if zeroed is not None else 0), int(x_act), float(x_act_param),
                                    int(LINEAR_PRECISION if precision is None else precision), _p(gz) if gz is not None else None,
                                    _p(gx) if gx is not None else None, _p(gw) if gw is not None else None,
                                    _p(gb) if gb is not None else None, _p(gbx) if gbx is not None else None,
                                    _p(n_dev, torch.int64) if n_dev is not None else None, _stream()), "tf_linear_bwd_fused")
    return gx, gw, gb, gbx


def pwquad(wv, y, inverse):
    """ElementWisePWQuadraticTransform.flow_inv (inverse=False: density direction) / .flow (inverse=True: sampling direction) on
    parameter rows wv [m,21] and y [m] -> out [m], logj [m], bins [m] int32 (flow.py:332-525; the device functions of the fused
    flow kernels)."""
    lib = L.load()
    wv, y = _f(wv), _f(y)
    m = y.shape[0]
    assert wv.shape == (m, 21)
    out, logj = torch.empty_like(y), torch.empty_like(y)
    bins = torch.empty(m, dtype=torch.int32, device=y.device)
    L.check(lib.tf_pwquad_eval(_p(wv), _p(y), m, 1 if inverse else 0, _p(out), _p(logj), _p(bins, torch.int32), _stream()), "tf_pwquad_eval")
    return out, logj, bins


def flow_logq(weights, cond, x, rays_id=None, want_bins=False, precision=1):
    """x [pn,sn,2] (rays_id None) or [m,2] with rays_id [m] int64 -> z (same shape), logq [...,1]."""
    lib = L.load()
    cond, x = _f(cond), _f(x)
    pn = cond.shape[0]
    dev = cond.device
    shape = x.shape[:-1]
    m = int(np.prod(shape)) if len(shape) else 1
    sn = x.shape[1] if rays_id is None else 1
    nets, keep = _coupling_nets(weights)
    z = torch.empty_like(x)
    lq = torch.empty(*shape, 1, dtype=torch.float32, device=dev)
    bins = torch.empty(*shape, 2, dtype=torch.int32, device=dev) if want_bins else None
    ws = _workspace("flow", lib.tf_flow_workspace_floats(pn), dev)
    rid = None if rays_id is None else rays_id.contiguous()
    L.check(lib.tf_flow_logq_fwd(C.byref(nets), _p(cond), _p(x), _p(rid, torch.int64), m, sn, pn, _p(z), _p(lq),
                                 _p(bins, torch.int32), int(precision), _p(ws), ws.numel(), _stream()), "tf_flow_logq_fwd")
    return (z, lq, bins) if want_bins else (z, lq)


def flow_logq_bwd(weights, cond, x, g_logq, rays_id=None, want_gx=False, z=None):
    """Backward of flow_logq wrt the 16 net tensors and cond (want_gx: and wrt the sample coordinates x, closed form in the kernel).
    z: the forward's z on the same inputs (the kernel then skips its first re-evaluation of block 1's net).
    -> (grads: 2 lists of 4 (gW, gb) pairs in torch layout, g_cond [pn,37]) (, g_x like x)."""
    lib = L.load()
    cond, x, g_logq = _f(cond), _f(x), _f(g_logq.reshape(-1))
    pn = cond.shape[0]
    dev = cond.device
    shape = x.shape[:-1]
    m = int(np.prod(shape)) if len(shape) else 1
    sn = x.shape[1] if rays_id is None else 1
    nets, keep = _coupling_nets(weights)
    gnets = (L.TfCouplingNetGrad * 2)()
    # every gradient buffer of the call -- 16 net tensors + the hoisted per-point rows -- is a view into ONE zero-filled allocation
    # (was 17 fills per call); 16-byte aligned slices
    sizes = [(tuple(W.shape), tuple(b.shape)) for k in range(2) for (W, b) in weights[k]]
    al = lambda n: (n + 3) // 4 * 4
    n_x = al(x.numel()) if want_gx else 0
    total = sum(al(int(np.prod(ws_))) + al(int(np.prod(bs_))) for ws_, bs_ in sizes) + 2 * pn * 64 + al(pn * 37) + n_x
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    grads, off = [[], []], 0
    for k in range(2):
        for l in range(4):
            ws_, bs_ = sizes[4 * k + l]
            nw, nb = int(np.prod(ws_)), int(np.prod(bs_))
            gw = flat[off:off + nw].view(ws_); off += al(nw)
            gb_ = flat[off:off + nb].view(bs_); off += al(nb)
            grads[k].append((gw, gb_))
            gnets[k].w[l], gnets[k].b[l] = gw.data_ptr(), gb_.data_ptr()
    g_point = flat[off:off + 2 * pn * 64].view(2, pn, 64); off += 2 * pn * 64
    g_cond = flat[off:off + pn * 37].view(pn, 37); off += al(pn * 37)
    g_x = flat[off:off + x.numel()].view(x.shape) if want_gx else None
    ws = _workspace("flow_bwd", lib.tf_flow_bwd_workspace_floats(pn), dev)
    rid = None if rays_id is None else rays_id.contiguous()
    if z is not None:
        z = _f(z)
        assert z.shape == x.shape
    # the hoisted per-point part is folded inside the call (g_cond given): d W1[:, 7:], d b1 and d cond are complete on return
    L.check(lib.tf_flow_logq_bwd(C.byref(nets), _p(cond), _p(x), _p(z), _p(rid, torch.int64), m, sn, pn, _p(g_logq), C.byref(gnets),
                                 _p(g_point), _p(g_cond), _p(g_x), _p(ws), ws.numel(), _stream()), "tf_flow_logq_bwd")
    return (grads, g_cond, g_x) if want_gx else (grads, g_cond)


# ------------------------------------------------------------------------------ light / mesh / shading
def cube_lookup(base, dirs, apply_exp=True, depth=None, near_eps=0.0):
    lib = L.load()
    base, dirs = _f(base), _f(dirs.reshape(-1, 3))
    assert base.dim() == 4 and base.shape[0] == 6 and base.shape[1] == base.shape[2] and base.shape[3] == 3
    out = torch.empty_like(dirs)
    L.check(lib.tf_cube_lookup_fwd(_p(base), base.shape[1], _p(dirs), dirs.shape[0], int(apply_exp),
                                   _p(None if depth is None else _f(depth.reshape(-1))), float(near_eps), _p(out), _stream()),
            "tf_cube_lookup_fwd")
    return out


def compact_mask(mask_u8):
    """-> idx [m] int64 (first *count entries valid, unordered), count [1] int64 -- both on the device, no sync."""
    lib = L.load()
    m = mask_u8.numel()
    idx = torch.empty(m, dtype=torch.int64, device=mask_u8.device)
    count = torch.empty(1, dtype=torch.int64, device=mask_u8.device)
    L.check(lib.tf_compact_mask(_p(mask_u8.reshape(-1).contiguous(), torch.uint8), m, _p(idx, torch.int64), _p(count, torch.int64),
                                _stream()), "tf_compact_mask")
    return idx, count


MISS_DEPTH = 10.0      # TF_MISS_DEPTH: tf_bvh_trace's depth of a ray that hits nothing


def compact_below(v, thr):
    """Indices i with v[i] < thr -> idx [m] int64 (first *count entries valid, unordered), count [1] int64 (tf_compact_below): the hit
    list straight from the traversal's depth array."""
    lib = L.load()
    v = _f(v).reshape(-1)
    m = v.numel()
    idx = torch.empty(m, dtype=torch.int64, device=v.device)
    count = torch.empty(1, dtype=torch.int64, device=v.device)
    L.check(lib.tf_compact_below(_p(v), float(thr), m, _p(idx, torch.int64), _p(count, torch.int64), _stream()), "tf_compact_below")
    return idx, count


def _mlp4(weights, in_dim=123):
    net = L.TfMlp4()
    keep = []
    expect = [(256, in_dim), (256, 256), (256, 256), (3, 256)]
    for l in range(4):
        W, b = _f(weights[l][0]), _f(weights[l][1])
        if tuple(W.shape) != expect[l]:
            raise RuntimeError(f"inner light layer {l}: weight shape {tuple(W.shape)} != {expect[l]}")
        keep += [W, b]
        net.w[l], net.b[l] = W.data_ptr(), b.data_ptr()
    return net, keep


def inner_light_indexed(weights, pos, dirs, nrm, idx, count, depth, lights, near_eps=1e-5, exp_max=5.0, precision=1, cache=None, acts=None):
    """In place: lights[i] = inner_light(pos[i], -dirs[i], nrm[i]) * (depth[i] > near_eps) for i in idx[:count].
    acts: [3, idx.numel(), 256] fp32 (training, precision PREC_F16X3): also receives the three hidden layers' post-ReLU activations,
    row r = ray idx[r] (tf_inner_light_indexed_train_fwd)."""
    lib = L.load()
    net, keep = _mlp4(weights)
    if cache is None:
        ws, flag = _workspace("inner", lib.tf_inner_light_workspace_floats(), pos.device), 0
    else:
        ws = cache.workspace(lib.tf_inner_light_workspace_floats(), pos.device)
        flag = cache.flag(keep, precision)
    if acts is not None:
        assert acts.is_contiguous() and acts.dtype == torch.float32 and tuple(acts.shape) == (3, idx.numel(), 256)
        L.check(lib.tf_inner_light_indexed_train_fwd(C.byref(net), _p(pos), _p(dirs), _p(nrm), _p(idx, torch.int64), _p(count, torch.int64),
                                                     idx.numel(), _p(depth), float(near_eps), float(exp_max), int(precision) | flag, _p(lights),
                                                     _p(acts), _p(ws), ws.numel(), _stream()), "tf_inner_light_indexed_train_fwd")
        return lights
    L.check(lib.tf_inner_light_indexed_fwd(C.byref(net), _p(pos), _p(dirs), _p(nrm), _p(idx, torch.int64), _p(count, torch.int64),
                                           idx.numel(), _p(depth), float(near_eps), float(exp_max), int(precision) | flag, _p(lights),
                                           _p(ws), ws.numel(), _stream()), "tf_inner_light_indexed_fwd")
    return lights


def outer_light_indexed(weights, dirs, idx, count, lights, exp_max=5.0, precision=PREC_F16X3, cache=None):
    """In place: lights[i] = exp(min(outer_light(IDE5(dirs[i])), exp_max)) for i in idx[:count] -- predict_outer_lights('direction') on
    the rays that missed (tf_outer_light_indexed_fwd).  weights: 4 (W_eff, b) pairs, 72-256-256-256-3."""
    lib = L.load()
    net, keep = _mlp4(weights, in_dim=72)
    if cache is None:
        ws, flag = _workspace("outer", lib.tf_inner_light_workspace_floats(), dirs.device), 0
    else:
        ws = cache.workspace(lib.tf_inner_light_workspace_floats(), dirs.device)
        flag = cache.flag(keep, precision)
    L.check(lib.tf_outer_light_indexed_fwd(C.byref(net), _p(dirs), _p(idx, torch.int64), _p(count, torch.int64), idx.numel(),
                                           float(exp_max), int(precision) | flag, _p(lights), _p(ws), ws.numel(), _stream()),
            "tf_outer_light_indexed_fwd")
    return lights


def cube_lookup_bwd(base, dirs, g_out, apply_exp=True):
    lib = L.load()
    base, dirs, g_out = _f(base), _f(dirs.reshape(-1, 3)), _f(g_out.reshape(-1, 3))
    g_base = torch.zeros_like(base)
    L.check(lib.tf_cube_lookup_bwd(_p(base), base.shape[1], _p(dirs), dirs.shape[0], int(apply_exp), _p(g_out), _p(g_base),
                                   _stream()), "tf_cube_lookup_bwd")
    return g_base


def _cube_stack(texs):
    texs = [_f(t) for t in texs]
    n = len(texs)
    if not 1 <= n <= 8:
        raise RuntimeError(f"cube_lookup_mips: 1..8 levels, got {n}")
    for t in texs:
        if t.dim() != 4 or t.shape[0] != 6 or t.shape[1] != t.shape[2] or t.shape[3] != 3:
            raise RuntimeError(f"cube_lookup_mips: every level is [6,R,R,3], got {tuple(t.shape)}")
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in texs])
    res = (C.c_int32 * n)(*[int(t.shape[1]) for t in texs])
    return texs, n, ptrs, res


def cube_lookup_mips(texs, dirs, mip, apply_exp=True):
    """tf_cube_lookup_mips_fwd: texs = the specular stack ([6,R_l,R_l,3] each), dirs [m,3], mip [m] -> [m,3]."""
    texs, n, ptrs, res = _cube_stack(texs)
    dirs, mip = _f(dirs), _f(mip.reshape(-1))
    m = dirs.shape[0]
    if dirs.shape != (m, 3) or mip.numel() != m:
        raise RuntimeError(f"cube_lookup_mips: dirs [m,3] and mip [m], got {tuple(dirs.shape)} and {mip.numel()} values")
    out = torch.empty(m, 3, dtype=torch.float32, device=dirs.device)
    L.check(L.load().tf_cube_lookup_mips_fwd(ptrs, res, n, _p(dirs), _p(mip), m, int(apply_exp), _p(out), _stream()), "tf_cube_lookup_mips_fwd")
    return out


def cube_lookup_mips_bwd(texs, dirs, mip, g_out, apply_exp=True, want_texs=True, want_dirs=True, want_mip=True):
    """-> (list of g_tex or None, g_dirs or None, g_mip or None)."""
    texs, n, ptrs, res = _cube_stack(texs)
    dirs, mip, g_out = _f(dirs), _f(mip.reshape(-1)), _f(g_out)
    m = dirs.shape[0]
    if dirs.shape != (m, 3) or mip.numel() != m or g_out.shape != (m, 3):
        raise RuntimeError(f"cube_lookup_mips_bwd: dirs / g_out [m,3] and mip [m], got {tuple(dirs.shape)}, {tuple(g_out.shape)}, {mip.numel()}")
    # ONE zero-filled buffer carved into the gradient arrays (the kernel accumulates with atomics): one fill launch instead of n + 2
    sizes = ([t.numel() for t in texs] if want_texs else []) + ([3 * m] if want_dirs else []) + ([m] if want_mip else [])
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dirs.device)
    parts = list(torch.split(flat, sizes)) if sizes else []
    g_texs = [parts.pop(0).view_as(t) for t in texs] if want_texs else None
    gptrs = (C.c_void_p * n)(*[t.data_ptr() for t in g_texs]) if want_texs else None
    g_dirs = parts.pop(0).view(m, 3) if want_dirs else None
    g_mip = parts.pop(0) if want_mip else None
    L.check(L.load().tf_cube_lookup_mips_bwd(ptrs, res, n, _p(dirs), _p(mip), m, int(apply_exp), _p(g_out), gptrs, _p(g_dirs), _p(g_mip),
                                             _stream()), "tf_cube_lookup_mips_bwd")
    return g_texs, g_dirs, g_mip


def cube_lookup_bwd_dirs(base, dirs, g_out, apply_exp=False, want_base=True):
    """-> (g_base [6,R,R,3] or None, g_dirs [m,3]): gradient of the bilinear cube fetch wrt the map and wrt the direction."""
    lib = L.load()
    base, dirs, g_out = _f(base), _f(dirs.reshape(-1, 3)), _f(g_out.reshape(-1, 3))
    g_base = torch.zeros_like(base) if want_base else None
    g_dirs = torch.empty_like(dirs)
    L.check(lib.tf_cube_lookup_bwd_dirs(_p(base), base.shape[1], _p(dirs), dirs.shape[0], int(apply_exp), _p(g_out), _p(g_base),
                                        _p(g_dirs), _stream()), "tf_cube_lookup_bwd_dirs")
    return g_base, g_dirs


# ------------------------------------------------------------------------------ march samplers
def alpha_mask_sample(volume_u8, aabb, pts):
    """AlphaGridMask.sample_alpha(pts) > 0 -> bool [n]; volume_u8 [D,H,W] in {0,1}."""
    assert volume_u8.dim() == 3
    pts = _f(pts)
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.uint8, device=pts.device)
    D, H, W = volume_u8.shape
    L.check(L.load().tf_alpha_mask_sample(_p(volume_u8, torch.uint8), D, H, W, C.byref(_aabb6(aabb)), _p(pts), n,
                                          _p(out, torch.uint8), _stream()), "tf_alpha_mask_sample")
    return out.bool()


def march_uniform(o, d, near, far, aabb, n_steps, step_size=0.0, volume_u8=None, mask_aabb=None, cells=False, t_jitter=None):
    """-> t_starts [N], t_ends [N], ray_indices [N] int64 packed by (ray, t).  One host read of N between the two passes.
    cells: volume_u8 is an occupancy grid [rx,ry,rz] looked up per cell (OccGridEstimator.binaries) instead of an AlphaGridMask
    volume; t_jitter [rn]: per-ray start offset (stratified sampling)."""
    lib = L.load()
    o, d, near, far = _f(o), _f(d), _f(near.reshape(-1)), _f(far.reshape(-1))
    rn = o.shape[0]
    dev = o.device
    box = _aabb6(aabb)
    mbox = _aabb6(aabb if mask_aabb is None else mask_aabb)
    D, H, W = (0, 0, 0) if volume_u8 is None else volume_u8.shape
    vol = _p(None if volume_u8 is None else volume_u8, torch.uint8)
    counts = torch.empty(rn, dtype=torch.int64, device=dev)
    null = C.c_void_p(0)
    mode = 1 if cells else 0
    tj = _p(_f(t_jitter.reshape(-1))) if t_jitter is not None else None
    L.check(lib.tf_march_uniform(_p(o), _p(d), _p(near), _p(far), rn, n_steps, float(step_size), C.byref(box), vol, D, H, W,
                                 C.byref(mbox), mode, tj, null, _p(counts, torch.int64), null, null, null, _stream()), "tf_march_uniform")
    incl = torch.cumsum(counts, 0)
    offsets = (incl - counts).contiguous()
    n = int(incl[-1]) if rn > 0 else 0
    t0 = torch.empty(n, dtype=torch.float32, device=dev)
    t1 = torch.empty(n, dtype=torch.float32, device=dev)
    ridx = torch.empty(n, dtype=torch.int64, device=dev)
    if n > 0:
        L.check(lib.tf_march_uniform(_p(o), _p(d), _p(near), _p(far), rn, n_steps, float(step_size), C.byref(box), vol, D, H, W,
                                     C.byref(mbox), mode, tj, _p(offsets, torch.int64), null, _p(t0), _p(t1), _p(ridx, torch.int64),
                                     _stream()), "tf_march_uniform")
    return t0, t1, ridx


# ------------------------------------------------------------------------------ env-light prefilter
def _cube(c):
    c = _f(c)
    assert c.dim() == 4 and c.shape[0] == 6 and c.shape[1] == c.shape[2] and c.shape[3] == 3, f"bad cube map {tuple(c.shape)}"
    return c


def cubemap_mip(cube):
    cube = _cube(cube)
    R = cube.shape[1]
    out = torch.empty(6, R // 2, R // 2, 3, dtype=torch.float32, device=cube.device)
    L.check(L.load().tf_cubemap_mip_fwd(_p(cube), R, _p(out), _stream()), "tf_cubemap_mip_fwd")
    return out


def cubemap_diffuse(cube, adjoint=False):
    cube = _cube(cube)
    out = torch.empty_like(cube)
    lib = L.load()
    fn = lib.tf_cubemap_diffuse_bwd if adjoint else lib.tf_cubemap_diffuse_fwd
    L.check(fn(_p(cube), cube.shape[1], _p(out), _stream()), "tf_cubemap_diffuse")
    return out


_TEXEL_TABLES = {}


def cubemap_texel_table(res, device):
    """[6,res,res,4] (unit direction, area) of every texel (tf_cubemap_texel_table): a function of the resolution alone, built once per
    (resolution, device) and handed to the GGX prefilter and its adjoint (round 5)."""
    key = (int(res), str(device))
    if key not in _TEXEL_TABLES:
        tab = torch.empty(6, res, res, 4, dtype=torch.float32, device=device)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("cubemap_texel_table: build the table (one call of EnvLight.build_mips) before capturing a graph")
        L.check(L.load().tf_cubemap_texel_table(int(res), _p(tab), _stream()), "tf_cubemap_texel_table")
        # the table is cached and later handed to kernels on ANY stream (side streams, shade_many's call streams, the autograd backward
        # stream): complete it here, once per (resolution, device), instead of ordering every consumer behind the builder's stream
        torch.cuda.current_stream(tab.device).synchronize()
        _TEXEL_TABLES[key] = tab
    return _TEXEL_TABLES[key]


def cubemap_specular(cube, roughness, cos_cutoff, use_table=True):
    """-> (filtered map, weight sums [6,R,R])"""
    cube = _cube(cube)
    out = torch.empty_like(cube)
    wsum = torch.empty(cube.shape[:3], dtype=torch.float32, device=cube.device)
    tab = cubemap_texel_table(cube.shape[1], cube.device) if use_table else None
    L.check(L.load().tf_cubemap_specular_fwd(_p(cube), cube.shape[1], float(roughness), float(cos_cutoff), _p(out), _p(wsum), _p(tab),
                                             _stream()), "tf_cubemap_specular_fwd")
    return out, wsum


def cubemap_specular_bwd(g_out, wsum, roughness, cos_cutoff, use_table=True):
    g_out = _cube(g_out)
    g = torch.empty_like(g_out)
    tab = cubemap_texel_table(g_out.shape[1], g_out.device) if use_table else None
    L.check(L.load().tf_cubemap_specular_bwd(_p(g_out), _p(_f(wsum)), g_out.shape[1], float(roughness), float(cos_cutoff), _p(g), _p(tab),
                                             _stream()), "tf_cubemap_specular_bwd")
    return g


class Bvh:
    """Host-built BVH uploaded to the device (replaces raytracing.RayTracer, raytracing/raytracer.py:7-17)."""

    def __init__(self, vertices, triangles, device="cuda"):
        self.lib = L.load()
        v = np.ascontiguousarray(np.asarray(vertices, dtype=np.float32))
        f = np.ascontiguousarray(np.asarray(triangles, dtype=np.int32))
        if f.shape[0] <= 8:
            raise AssertionError("BVH needs at least 8 triangles.")       # raytracer.py:15
        nodes = np.zeros((2 * f.shape[0], 8), dtype=np.float32)
        tris = np.zeros((f.shape[0], 9), dtype=np.float32)
        n = self.lib.tf_bvh_build_host(v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], nodes.ctypes.data, tris.ctypes.data)
        if n <= 0:
            L.check(int(n), "tf_bvh_build_host")
        self.n_nodes = int(n)
        rec = int(self.lib.tf_bvh_record_dwords())          # 8: child pairs; 16: 4-wide nodes (a build-time choice of the library)
        pairs = np.zeros((n // 2 + 1, rec), dtype=np.uint32)
        tris12 = np.zeros((f.shape[0], 12), dtype=np.float32)
        self.frame = (C.c_float * 6)()
        npair = self.lib.tf_bvh_pack_host(nodes.ctypes.data, int(n), tris.ctypes.data, f.shape[0], pairs.ctypes.data, tris12.ctypes.data,
                                          C.addressof(self.frame))
        if npair <= 0:
            L.check(int(npair), "tf_bvh_pack_host")
        self.n_pairs = int(npair)
        self.pairs = torch.from_numpy(pairs[:npair].view(np.int32).copy()).to(device)
        self.tris = torch.from_numpy(tris12).to(device)

    def trace(self, o, d, off0=0.0, off1=0.0, want_pos=True, want_nrm=True, live=None, dynamic=True, slot_order=None,
              hit_rows_only=False, origin_order=None, want_hit=True):
        """o [m,3] (one origin per ray) or [m // T, 3] (T consecutive rays share an origin row); d [m,3].
        hit_rows_only: the pos / nrm rows of rays that miss are left UNINITIALISED (callers that only read hit rows).
        origin_order [n_origins] int32: order in which the origins are handed to the persistent waves (see morton_order)."""
        o, d = _f(o.reshape(-1, 3)), _f(d.reshape(-1, 3))
        m = d.shape[0]
        if m == 0:
            per_origin = 1
        elif o.shape[0] == 0 or m % o.shape[0] != 0:
            raise RuntimeError(f"Bvh.trace: {m} directions cannot share {o.shape[0]} origins")
        else:
            per_origin = m // o.shape[0]
        if slot_order is not None and (slot_order.dtype != torch.int32 or slot_order.numel() != per_origin):
            raise RuntimeError("Bvh.trace: slot_order must be an int32 permutation of the rays of one origin")
        dev = o.device
        pos = torch.empty(m, 3, dtype=torch.float32, device=dev) if want_pos else None
        nrm = torch.empty(m, 3, dtype=torch.float32, device=dev) if want_nrm else None
        depth = torch.empty(m, dtype=torch.float32, device=dev)
        hit = torch.empty(m, dtype=torch.uint8, device=dev) if want_hit else None       # want_hit=False: a ray hit iff depth < 10 (MISS_DEPTH)
        lv = None if live is None else live.reshape(-1).contiguous()
        ctr = torch.empty(8, dtype=torch.int64, device=dev) if dynamic else None
        if origin_order is not None and (origin_order.dtype != torch.int32 or origin_order.numel() != o.shape[0]):
            raise RuntimeError("Bvh.trace: origin_order must be an int32 permutation of the origin rows")
        L.check(self.lib.tf_bvh_trace(_p(self.pairs, torch.int32), _p(self.tris), C.byref(self.frame), self.n_pairs, _p(o), _p(d), per_origin, _p(slot_order, torch.int32), float(off0), float(off1),
                                      _p(lv, torch.uint8), m, _p(pos), _p(nrm), _p(depth), _p(hit, torch.uint8),
                                      int(bool(hit_rows_only)), _p(origin_order, torch.int32), _p(ctr, torch.int64), _stream()), "tf_bvh_trace")
        return pos, nrm, depth, hit.bool() if want_hit else None


def morton_order(pts, aabb):
    """int32 permutation sorting points [n,3] along a 30-bit Morton curve over `aabb` (device-side; Bvh.trace origin_order)."""
    lo, hi = aabb[0].to(pts.device), aabb[1].to(pts.device)
    q = ((pts - lo) / (hi - lo)).clamp(0, 1).mul(1023).long()
    code = torch.zeros(pts.shape[0], dtype=torch.long, device=pts.device)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code).to(torch.int32)


class ShapeShade:
    """Fused split-sum shading of the shape stage (tf_shape_shade_pack / tf_shape_shade_fwd).
    nets: {"mat_mlp" | "inner_light" | "inner_weight": [(W, b)] * 3} (weight-norm folded); spec: list of [6,R,R,3] pre-filtered
    specular mips, diff [6,Rd,Rd,3], fg_lut [1,H,W,2] or [H,W,2]."""

    def __init__(self, nets, spec, diff, fg_lut, min_roughness=0.08, max_roughness=0.5, light_exp_max=0.0):
        self.lib = L.load()
        dev = spec[0].device
        self.spec = [_f(t) for t in spec]
        self.diff = _f(diff)
        self.fg = _f(fg_lut.reshape(fg_lut.shape[-3], fg_lut.shape[-2], 2))
        for t in self.spec + [self.diff, self.fg]:
            _p(t)
        self.spec_ptrs = (C.c_void_p * len(self.spec))(*[t.data_ptr() for t in self.spec])
        self.spec_res = (C.c_int32 * len(self.spec))(*[t.shape[1] for t in self.spec])
        self.consts = (float(min_roughness), float(max_roughness), float(light_exp_max))
        self.ws = torch.empty(int(self.lib.tf_shape_shade_workspace_floats()), dtype=torch.float32, device=dev)
        self.repack(nets)

    def repack(self, nets):
        sn = L.TfShapeNets()
        keep = []
        expect = {"mat_mlp": [(128, 128), (128, 128), (5, 128)], "inner_light": [(128, 123), (128, 128), (3, 128)],
                  "inner_weight": [(128, 90), (128, 128), (1, 128)]}
        for name in ("mat_mlp", "inner_light", "inner_weight"):
            m = getattr(sn, name)
            for l in range(3):
                W, b = _f(nets[name][l][0]), _f(nets[name][l][1])
                if tuple(W.shape) != expect[name][l]:
                    raise RuntimeError(f"ShapeShade: {name} layer {l} has shape {tuple(W.shape)}, expected {expect[name][l]}")
                _p(W), _p(b)
                keep += [W, b]
                m.w[l], m.b[l] = W.data_ptr(), b.data_ptr()
        L.check(self.lib.tf_shape_shade_pack(C.byref(sn), _p(self.ws), self.ws.numel(), _stream()), "tf_shape_shade_pack")

    def __call__(self, pts, normals, view, feat):
        """-> color [n,3], occ [n,1], roughness [n,1], refl [n,3]"""
        pts, normals, view, feat = _f(pts), _f(normals), _f(view), _f(feat)
        n, dev = pts.shape[0], pts.device
        if feat.shape != (n, 128):
            raise RuntimeError(f"ShapeShade: feat must be [n,128], got {tuple(feat.shape)}")
        color = torch.empty(n, 3, dtype=torch.float32, device=dev)
        occ = torch.empty(n, 1, dtype=torch.float32, device=dev)
        rough = torch.empty(n, 1, dtype=torch.float32, device=dev)
        refl = torch.empty(n, 3, dtype=torch.float32, device=dev)
        L.check(self.lib.tf_shape_shade_fwd(_p(self.ws), C.addressof(self.spec_ptrs), C.addressof(self.spec_res), len(self.spec),
                                            _p(self.diff), self.diff.shape[1], _p(self.fg), self.fg.shape[0], self.fg.shape[1],
                                            *self.consts, _p(pts), _p(normals), _p(view), _p(feat), n, _p(color), _p(occ), _p(rough),
                                            _p(refl), _stream()), "tf_shape_shade_fwd")
        return color, occ, rough, refl


class PointPrep:
    """Fused per-point stage (tf_point_pack / tf_point_fwd): materials + both flow condition rows in one launch.
    mat_nets: {"metallic" | "roughness" | "albedo": [(W1 [128,108], b1), (W2, b2)]} (weight-norm folded);
    nis_nets: [diffuse, specular] each [(W1 [64,57], b1), (W2 [16,64], b2)]."""

    def __init__(self, mat_packed: VmPacked, flow_d_packed: VmPacked, flow_s_packed: VmPacked, mat_nets, nis_nets, aabb,
                 rough_min=0.04):
        self.lib = L.load()
        self.fields = (mat_packed, flow_d_packed, flow_s_packed)
        self.aabb6 = _aabb6(aabb)
        self.rough_min = float(rough_min)
        dev = (mat_packed.data16 if mat_packed.texel_f16 else mat_packed.data).device
        self.ws = torch.empty(int(self.lib.tf_point_workspace_floats()), dtype=torch.float32, device=dev)
        self.repack(mat_nets, nis_nets)

    def repack(self, mat_nets, nis_nets):
        nets = L.TfPointNets()
        keep = []
        for n, name in enumerate(("metallic", "roughness", "albedo")):
            (w1, b1), (w2, b2) = mat_nets[name]
            t = [_f(w1), _f(b1), _f(w2), _f(b2)]
            if tuple(t[0].shape) != (128, 108) or t[2].shape[1] != 128 or t[2].shape[0] != (3 if name == "albedo" else 1):
                raise RuntimeError(f"PointPrep: {name} predictor has shapes {tuple(t[0].shape)}, {tuple(t[2].shape)}")
            keep += t
            nets.mat_w1[n], nets.mat_b1[n], nets.mat_w2[n], nets.mat_b2[n] = [x.data_ptr() for x in t]
        for f in range(2):
            (w1, b1), (w2, b2) = nis_nets[f]
            t = [_f(w1), _f(b1), _f(w2), _f(b2)]
            if tuple(t[0].shape) != (64, 57) or tuple(t[2].shape) != (16, 64):
                raise RuntimeError(f"PointPrep: flow feature net {f} has shapes {tuple(t[0].shape)}, {tuple(t[2].shape)}")
            keep += t
            nets.nis_w1[f], nets.nis_b1[f], nets.nis_w2[f], nets.nis_b2[f] = [x.data_ptr() for x in t]
        for t in keep:
            _p(t)
        L.check(self.lib.tf_point_pack(C.byref(nets), _p(self.ws), self.ws.numel(), _stream()), "tf_point_pack")

    def __call__(self, pts, view_angles):
        """-> metallic [pn,1], roughness [pn,1], albedo [pn,3], cond_d [pn,37], cond_s [pn,37]"""
        pts, va = _f(pts), _f(view_angles)
        pn, dev = pts.shape[0], pts.device
        met = torch.empty(pn, 1, dtype=torch.float32, device=dev)
        rough = torch.empty(pn, 1, dtype=torch.float32, device=dev)
        alb = torch.empty(pn, 3, dtype=torch.float32, device=dev)
        cd = torch.empty(pn, 37, dtype=torch.float32, device=dev)
        cs = torch.empty(pn, 37, dtype=torch.float32, device=dev)
        m, fd, fs = self.fields
        L.check(self.lib.tf_point_fwd(_p(self.ws), C.byref(m.desc), m.ptr(), C.byref(fd.desc), fd.ptr(), C.byref(fs.desc),
                                      fs.ptr(), C.byref(self.aabb6), _p(pts), _p(va), pn, self.rough_min, _p(met), _p(rough),
                                      _p(alb), _p(cd), _p(cs), _stream()), "tf_point_fwd")
        return met, rough, alb, cd, cs


WEIGHTS_PACKED = 0x100


class PackCache:
    """Caller-side memo for ONE network's packed-weight workspace: repeated calls with unchanged weights pass
    TF_WEIGHTS_PACKED and skip the fragment re-pack launches (the library itself is stateless).  Weight updates are
    detected through (data_ptr, torch _version) of every weight tensor.  One workspace PER STREAM the network is called on: a
    workspace also holds per-call rows (the flow kernels' hoisted per-point layer-1 part), so two calls in flight on two streams
    must not share one (round 4 measured exactly that as run-to-run different checksums)."""

    def __init__(self):
        self._per_stream = {}

    def _slot(self):
        return self._per_stream.setdefault(int(torch.cuda.current_stream().cuda_stream), {"ws": None, "key": None})

    @property
    def ws(self):
        return self._slot()["ws"]

    def workspace(self, n_floats, device):
        sl = self._slot()
        if sl["ws"] is None or sl["ws"].numel() < n_floats or sl["ws"].device != torch.device(device):
            sl["ws"] = torch.empty(int(n_floats), dtype=torch.float32, device=device)
            sl["key"] = None
        return sl["ws"]

    def flag(self, tensors, precision):
        sl = self._slot()
        key = (int(precision), tuple((t.data_ptr(), t._version) for t in tensors))
        hit = key == sl["key"]
        sl["key"] = key
        return WEIGHTS_PACKED if hit else 0


def inner_light(weights, pts, view, nrm, exp_max=5.0, precision=PREC_F16X3):
    """weights: 4 (W_eff, b) pairs, 123-256-256-256-3.  precision: PREC_F32 (exact fp32 MFMA) or PREC_F16X3."""
    lib = L.load()
    pts, view, nrm = _f(pts), _f(view), _f(nrm)
    net = L.TfMlp4()
    keep = []
    expect = [(256, 123), (256, 256), (256, 256), (3, 256)]
    for l in range(4):
        W, b = _f(weights[l][0]), _f(weights[l][1])
        if tuple(W.shape) != expect[l]:
            raise RuntimeError(f"inner light layer {l}: weight shape {tuple(W.shape)} != {expect[l]}")
        keep += [W, b]
        net.w[l], net.b[l] = W.data_ptr(), b.data_ptr()
    out = torch.empty_like(pts)
    ws = _workspace("inner", lib.tf_inner_light_workspace_floats(), pts.device)
    L.check(lib.tf_inner_light_fwd(C.byref(net), _p(pts), _p(view), _p(nrm), pts.shape[0], float(exp_max), int(precision), _p(out),
                                   _p(ws), ws.numel(), _stream()), "tf_inner_light_fwd")
    return out


def view_angles(normals, view):
    lib = L.load()
    normals, view = _f(normals), _f(view)
    va = torch.empty(normals.shape[0], 2, dtype=torch.float32, device=normals.device)
    L.check(lib.tf_view_angles(_p(normals), _p(view), normals.shape[0], _p(va), _stream()), "tf_view_angles")
    return va


def shade_dirs(normals, view, metallic, roughness, albedo, ang_d, logq_d, fixed_d, ang_s, logq_s, az_jitter=None, want_logjac=False,
               slot_of_pos=None, rows=None, out=None, whole=(False, False), smith=False):
    """whole = (diffuse, specular): that lobe's flow samples are OUTGOING directions, not half vectors (cfg use_half_* = False);
    smith: cfg geometry_type = 'ggx_smith' (the Smith geometry term instead of the Schlick-GGX product).
    slot_of_pos [T] int32 permutation: row j of dirs / wgt / live holds slot slot_of_pos[j] (a point's rays stored in traversal order).
    rows = (begin, count): build only these rows of every point, into the arrays `out` = (dirs, wgt, mask, live) of an earlier call
    (None: allocate); the sample arrays of direction sets outside the range are not read, so a set's rows can be built while the
    next set is still being sampled (ang_s / logq_s may then be given as shapes: (ss,))."""
    lib = L.load()
    pn = normals.shape[0]
    sd = 0 if ang_d is None else ang_d.shape[1]
    nf = 0 if fixed_d is None else fixed_d.shape[0]
    ss_only = isinstance(ang_s, tuple)                      # placeholder: the specular samples do not exist yet
    ss = 0 if ang_s is None else (ang_s[0] if ss_only else ang_s.shape[1])
    T = sd + nf + ss
    dev = normals.device
    if out is None:
        dirs = torch.empty(pn, T, 3, dtype=torch.float32, device=dev)
        wgt = torch.empty(pn, T, 3, dtype=torch.float32, device=dev)
        mask = torch.empty(pn, ss, dtype=torch.uint8, device=dev)
        live = torch.empty(pn, T, dtype=torch.uint8, device=dev)
    else:
        dirs, wgt, mask, live = out
        mask = mask.view(torch.uint8)
    logjac = torch.empty(pn, sd + ss, dtype=torch.float32, device=dev) if want_logjac else None
    g = lambda t: None if t is None else _f(t)
    if ss_only:
        a_s = l_s = dirs                                    # any valid address: never read for rows outside the specular set
    else:
        a_s, l_s = g(ang_s), g(None if logq_s is None else logq_s.reshape(pn, ss))
    r0, rc = (0, -1) if rows is None else (int(rows[0]), int(rows[1]))
    wm = (1 if whole[0] else 0) | (2 if whole[1] else 0) | (4 if smith else 0)
    L.check(lib.tf_shade_dirs_whole(_p(_f(normals)), _p(_f(view)), _p(_f(metallic.reshape(-1))), _p(_f(roughness.reshape(-1))),
                                    _p(_f(albedo)), _p(g(ang_d)), _p(g(None if logq_d is None else logq_d.reshape(pn, sd))), sd,
                                    _p(g(fixed_d)), _p(g(az_jitter)), nf, _p(a_s), _p(l_s), ss, pn, _p(dirs), _p(wgt),
                                    _p(mask, torch.uint8), _p(live, torch.uint8), _p(logjac), _p(slot_of_pos, torch.int32), r0, rc, wm,
                                    _stream()), "tf_shade_dirs")
    if want_logjac:
        return dirs, wgt, mask.bool(), live, logjac
    return dirs, wgt, mask.bool() if out is None else mask, live


def shade_dirs_fixed(normals, view, metallic, roughness, albedo, fixed_d, fixed_s, az_jitter=None, az_jitter_s=None, smith=False):
    """Direction sets of the non-NIS pass of shade_mixed: nf fixed cosine + ss fixed GGX-warped directions per point.
    -> dirs [pn,nf+ss,3], wgt [pn,nf+ss,3], spec_mask [pn,ss] bool, live [pn,nf+ss] u8."""
    lib = L.load()
    pn, nf, ss = normals.shape[0], fixed_d.shape[0], fixed_s.shape[0]
    dev = normals.device
    dirs = torch.empty(pn, nf + ss, 3, dtype=torch.float32, device=dev)
    wgt = torch.empty(pn, nf + ss, 3, dtype=torch.float32, device=dev)
    mask = torch.empty(pn, ss, dtype=torch.uint8, device=dev)
    live = torch.empty(pn, nf + ss, dtype=torch.uint8, device=dev)
    g = lambda t: None if t is None else _f(t)
    L.check(lib.tf_shade_dirs_fixed_mode(_p(_f(normals)), _p(_f(view)), _p(_f(metallic.reshape(-1))), _p(_f(roughness.reshape(-1))),
                                         _p(_f(albedo)), _p(_f(fixed_d)), _p(g(az_jitter)), nf, _p(_f(fixed_s)), _p(g(az_jitter_s)), ss, pn,
                                         _p(dirs), _p(wgt), _p(mask, torch.uint8), _p(live, torch.uint8), 4 if smith else 0, _stream()),
            "tf_shade_dirs_fixed_mode")
    return dirs, wgt, mask.bool(), live


def shade_dirs_bwd(normals, view, metallic, roughness, albedo, dirs, wgt, g_wgt, sd, nf, ss, smith=False):
    lib = L.load()
    pn = normals.shape[0]
    dev = normals.device
    g_alb = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    g_met = torch.empty(pn, dtype=torch.float32, device=dev)
    g_rough = torch.empty(pn, dtype=torch.float32, device=dev)
    L.check(lib.tf_shade_dirs_bwd_mode(_p(_f(normals)), _p(_f(view)), _p(_f(metallic.reshape(-1))), _p(_f(roughness.reshape(-1))),
                                       _p(_f(albedo)), _p(_f(dirs)), _p(_f(wgt)), _p(_f(g_wgt)), sd, nf, ss, pn, _p(g_alb), _p(g_met),
                                       _p(g_rough), 4 if smith else 0, _stream()), "tf_shade_dirs_bwd_mode")
    return g_alb, g_met, g_rough


def inner_light_encode(pos, dirs, nrm, idx=None, count=None, ld=123):
    """-> X [capacity,ld] (rows >= *count are undefined; columns >= 123 zero)."""
    lib = L.load()
    cap = pos.shape[0] if idx is None else idx.numel()
    X = torch.empty(cap, ld, dtype=torch.float32, device=pos.device)
    ws = _workspace("inner", lib.tf_inner_light_workspace_floats(), pos.device)
    L.check(lib.tf_inner_light_encode(_p(_f(pos)), _p(_f(dirs)), _p(_f(nrm)), _p(idx, torch.int64), _p(count, torch.int64), cap, _p(X), int(ld),
                                      _p(ws), ws.numel(), _stream()), "tf_inner_light_encode")
    return X


def shade_reduce_env(wgt, dirs, depth, hit_u8, hit_lights, env_base, n_diffuse, ss, near_eps=1e-5, slot_of_pos=None):
    """colors / diffuse / specular sums with the miss branch of get_lights evaluated inside the reduction."""
    lib = L.load()
    pn = wgt.shape[0]
    dev = wgt.device
    colors = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    dl = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    sl = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    env_base = _f(env_base) if env_base is not None else None       # None: every light is in hit_lights (outer_light_version='direction')
    L.check(lib.tf_shade_reduce_env(_p(_f(wgt)), _p(_f(dirs)), _p(_f(depth)), _p(hit_u8, torch.uint8) if hit_u8 is not None else None,
                                    _p(_f(hit_lights)), _p(env_base) if env_base is not None else None,
                                    env_base.shape[1] if env_base is not None else 0, float(near_eps), pn, n_diffuse, ss, _p(colors), _p(dl), _p(sl),
                                    _p(slot_of_pos, torch.int32), _stream()),
            "tf_shade_reduce_env")
    return colors, dl, sl


def shade_reduce_aux(wgt, spec_mask, n_diffuse, ss, lights=None, dirs=None, depth=None, hit_u8=None, hit_lights=None, env_base=None,
                     near_eps=1e-5, slot_of_pos=None, want_colors=True):
    """tf_shade_reduce_aux: the reduction plus the per-point statistics behind the rest of shade_mixed's output dict
    -> (colors, diffuse_lin, specular_lin, aux [pn,16]).  Either `lights` [pn,T,3] (and hit flags from hit_u8 / depth) or the
    (dirs, depth, hit_lights, env_base) set of shade_reduce_env."""
    lib = L.load()
    pn = wgt.shape[0]
    dev = wgt.device
    colors = torch.empty(pn, 3, dtype=torch.float32, device=dev) if want_colors else None
    dl = torch.empty(pn, 3, dtype=torch.float32, device=dev) if want_colors else None
    sl = torch.empty(pn, 3, dtype=torch.float32, device=dev) if want_colors else None
    aux = torch.empty(pn, 16, dtype=torch.float32, device=dev)
    env_base = _f(env_base) if env_base is not None else None
    sm = spec_mask.view(torch.uint8) if spec_mask.dtype == torch.bool else spec_mask
    L.check(lib.tf_shade_reduce_aux(_p(_f(wgt)), _p(_f(lights)) if lights is not None else None, _p(_f(dirs)) if dirs is not None else None,
                                    _p(_f(depth)) if depth is not None else None, _p(hit_u8, torch.uint8) if hit_u8 is not None else None,
                                    _p(_f(hit_lights)) if hit_lights is not None else None, _p(env_base) if env_base is not None else None,
                                    env_base.shape[1] if env_base is not None else 0, float(near_eps), _p(sm.contiguous(), torch.uint8), pn,
                                    n_diffuse, ss, _p(colors), _p(dl), _p(sl), _p(aux), _p(slot_of_pos, torch.int32), _stream()),
            "tf_shade_reduce_aux")
    return colors, dl, sl, aux


def shade_reduce(wgt, lights, n_diffuse, ss):
    lib = L.load()
    pn = wgt.shape[0]
    dev = wgt.device
    colors = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    dl = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    sl = torch.empty(pn, 3, dtype=torch.float32, device=dev)
    L.check(lib.tf_shade_reduce(_p(_f(wgt)), _p(_f(lights)), pn, n_diffuse, ss, _p(colors), _p(dl), _p(sl), _stream()),
            "tf_shade_reduce")
    return colors, dl, sl
