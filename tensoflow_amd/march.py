"""Host-side composition of the NeuS-style ray-march on the MI355X.

Mirrors ShapeRenderer.near_far_from_sphere (network/shapeRenderer.py:676-684), compute_ball_radii (:966-970),
sample_ray + upsample + cat_z_vals (:820-932, with utils/network_utils.py:117-147 sample_pdf) and the train branch
of render_core (:1105-1277).  All field evaluations (64 + 3x16 per ray in the sampler, 7 per live sample in
render_core) and the compositing scan run in libtensoflow_hip.so (tf_sdf_forward, tf_sdf_alpha_fwd,
tf_composite_fwd); the sampler's per-ray bookkeeping (sort / searchsorted / cumprod over <= 128 entries) is
device-resident torch ops on the same stream -- no host round trip.
"""
import torch
import torch.nn.functional as F

from . import ops


def near_far_from_sphere(o, d, radius=1.0):
    a = (d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (o * d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return (mid - radius).clamp(min=1e-3), mid + radius


def ball_radii(t, radii, cos):
    inv = 1.0 / cos
    tmp = (inv * inv - 1).sqrt() - radii
    return t * radii * cos / (tmp * tmp + 1.0).sqrt()


class SdfField:
    """TensoSDF parameters resident on the device + packed pyramid (rebuilt by `refresh()` after an optimizer step)."""

    def __init__(self, sd, aabb, grid_size, n_levels, device="cuda", prefix="sdf_network.", field_f16=False):
        g = lambda k: sd[prefix + k].to(device).float().contiguous()
        self.planes = [g(f"sdf_plane.{i}") for i in range(3)]
        self.lines = [g(f"sdf_line.{i}") for i in range(3)]
        self.W = [g("sdf_mat.0.weight"), g("sdf_mat.0.bias"), g("sdf_mat.2.weight"), g("sdf_mat.2.bias")]
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).cpu()
        self.aabb_dev = self.aabb.to(device)
        self.grid_size = torch.as_tensor(grid_size, dtype=torch.float32)
        self.units = ((self.aabb[1] - self.aabb[0]) / (self.grid_size - 1)).tolist()
        self.n_levels = n_levels
        self.device = device
        self.packed = ops.VmPacked(self.planes, self.lines, n_levels, texel_f16=field_f16)     # field_f16: half texels (configs[4])

    def refresh(self):
        self.packed.repack(self.planes, self.lines)

    def sdf(self, pts, level=None):
        return ops.sdf_forward(self.packed, *self.W, pts, level, self.aabb, want_feat=False)[0]

    def forward(self, pts, level=None):
        return ops.sdf_forward(self.packed, *self.W, pts, level, self.aabb, want_feat=True)

    def sdf_alpha(self, pts, level, dists, dirs, inv_s, cos_anneal, want_feat=True, want_hess=True, precision=None):
        return ops.sdf_alpha(self.packed, *self.W, pts, level, dists, dirs, self.aabb, self.units, inv_s, cos_anneal,
                             want_feat=want_feat, want_hess=want_hess, precision=precision)


def _sample_pdf_det(bins, weights, n):
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.5 / n, 1.0 - 0.5 / n, steps=n, device=bins.device).expand(list(cdf.shape[:-1]) + [n]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def _upsample(o, d, z, sdf, n_imp, inv_s):
    pts = o[:, None] + d[:, None] * z[..., None]
    rad = pts.norm(dim=-1)
    inside = (rad[:, :-1] < 1.0) | (rad[:, 1:] < 1.0)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    pz, nz = z[:, :-1], z[:, 1:]
    mid = (ps + ns) * 0.5
    cos = (ns - ps) / (nz - pz + 1e-5)
    prev = torch.cat([torch.zeros(z.shape[0], 1, device=z.device), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clip(-1e3, 0.0) * inside
    dist = nz - pz
    pc = torch.sigmoid((mid - cos * dist * 0.5) * inv_s)
    nc = torch.sigmoid((mid + cos * dist * 0.5) * inv_s)
    alpha = (pc - nc + 1e-5) / (pc + 1e-5)
    w = alpha * torch.cumprod(torch.cat([torch.ones(z.shape[0], 1, device=z.device), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return _sample_pdf_det(z, w, n_imp)


@torch.no_grad()
def sample_ray(field: SdfField, o, d, near, far, radiis, rays_cos, base_radii, n_samples=64, n_importance=64, up_steps=4,
               t_rand=None, inv_s_cap=None):
    """ShapeRenderer.sample_ray -> packed t_starts, t_ends, ray_indices (int64).  t_rand [rn,1] in [-0.5, 0.5): the per-ray
    stratification offset of perturb > 0 (shapeRenderer.py:888-890); inv_s_cap: clip_sample_variance (:905-907).
    Round 4: the slab test + uniform samples, every up-sampling round (NeuS weights, inverse-CDF resampling) and the stable merge
    are HIP kernels (tf_sample_ray_init / _upsample / _merge: ~30 launches instead of ~420); `sample_ray_torch` is the torch
    composition of rounds 1-3, kept as the checker of tests/test_gpu_march.py."""
    n_imp = n_importance // max(up_steps, 1)
    if not (o.is_cuda and 2 <= n_samples and n_samples + n_imp * max(up_steps - 1, 0) <= 128 and 1 <= n_imp <= 32):
        return sample_ray_torch(field, o, d, near, far, radiis, rays_cos, base_radii, n_samples, n_importance, up_steps, t_rand, inv_s_cap)
    rn = o.shape[0]
    t, pts, lv = ops.sample_ray_init(o, d, near, far, radiis, rays_cos, field.aabb, n_samples, base_radii, t_rand)
    sdf = field.sdf(pts, lv).reshape(rn, n_samples)
    for i in range(up_steps):
        inv_s = float(64 * 2 ** i if inv_s_cap is None else min(inv_s_cap, 64 * 2 ** i))
        last = i + 1 == up_steps
        new_t, npts, nlv = ops.sample_ray_upsample(o, d, radiis, rays_cos, t, sdf, n_imp, inv_s, base_radii, want_pts=not last)
        nsdf = None if last else field.sdf(npts, nlv).reshape(rn, n_imp)
        t, sdf2 = ops.sample_ray_merge(t, sdf, new_t, nsdf)
        sdf = sdf2 if sdf2 is not None else sdf
    # intervals and the inside-the-box test in one launch (round 5; was twelve), then ONE compaction (one host sync for the
    # data-dependent size); row-major order = the reference's t[inner] order, ray index = flat index // samples per ray
    t0_all, t1_all, inner = ops.sample_ray_intervals(o, d, t, field.aabb)
    keep = torch.nonzero(inner)[:, 0]
    return t0_all.index_select(0, keep), t1_all.index_select(0, keep), torch.div(keep, t.shape[1], rounding_mode="floor")


@torch.no_grad()
def sample_ray_torch(field: SdfField, o, d, near, far, radiis, rays_cos, base_radii, n_samples=64, n_importance=64, up_steps=4,
                     t_rand=None, inv_s_cap=None):
    """The torch composition of sample_ray (rounds 1-3): same function, ~420 launches; the checker of the kernels above."""
    rn = o.shape[0]
    dev = o.device
    aabb = field.aabb_dev
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    ra, rb = (aabb[1] - o) / vec, (aabb[0] - o) / vec
    tmin = torch.minimum(ra, rb).amax(-1).clamp(min=near[:, 0], max=far[:, 0])[:, None]
    tmax = torch.maximum(ra, rb).amin(-1).clamp(min=near[:, 0], max=far[:, 0])[:, None]
    t = tmin + (tmax - tmin) * torch.linspace(0.0, 1.0, n_samples, device=dev)[None]
    if t_rand is not None:
        t = t + t_rand * 2.0 / n_samples
    pts = o[:, None] + d[:, None] * t[..., None]
    lv = torch.log2(ball_radii(t[..., None], radiis[:, None], rays_cos[:, None]) / base_radii)
    sdf = field.sdf(pts.reshape(-1, 3), lv.reshape(-1)).reshape(rn, n_samples)
    for i in range(up_steps):
        inv_s = torch.ones(rn, t.shape[1] - 1, device=dev) * (64 * 2 ** i if inv_s_cap is None else min(inv_s_cap, 64 * 2 ** i))
        new_t = _upsample(o, d, t, sdf, n_importance // up_steps, inv_s)
        t_all, index = torch.sort(torch.cat([t, new_t], -1), -1)
        if i + 1 < up_steps:
            npts = o[:, None] + d[:, None] * new_t[..., None]
            nlv = torch.log2(ball_radii(new_t[..., None], radiis[:, None], rays_cos[:, None]) / base_radii)
            nsdf = field.sdf(npts.reshape(-1, 3), nlv.reshape(-1)).reshape(rn, -1)
            sdf = torch.gather(torch.cat([sdf, nsdf], -1), -1, index)
        t = t_all
    dists = t[:, 1:] - t[:, :-1]
    dists = torch.cat([dists, dists[:, -1:]], -1)
    mid = t + dists * 0.5
    ridx = torch.arange(rn, device=dev)[:, None].expand(rn, t.shape[1])
    p = o[:, None] + d[:, None] * mid[..., None]
    inner = ~((aabb[0] > p) | (p > aabb[1])).any(-1)
    return t[inner], (t + dists)[inner], ridx[inner]


class AlphaMask:
    """AlphaGridMask (shapeRenderer.py:78-97): binary occupancy volume [D,H,W] (u8 on the device) over `aabb`."""

    def __init__(self, aabb, volume):
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).cpu()
        self.volume = (volume > 0).to(torch.uint8).contiguous()

    def alive(self, pts):
        return ops.alpha_mask_sample(self.volume, self.aabb, pts)


@torch.no_grad()
def update_alpha_mask(field: SdfField, inv_s, grid=(128, 128, 128), thres=1e-4, mul_length=10, prev: AlphaMask = None, chunk=1 << 21,
                      return_alpha=False):
    """ShapeRenderer.updateAlphaMask + compute_gridAlpha + compute_grid_alpha (shapeRenderer.py:257-325): NeuS opacity of one
    grid step at every lattice point (forced to 1 within mul_length steps of the surface), 3^3 max-pool dilation, threshold.
    Returns (AlphaMask, new_aabb [2,3]).  The field evaluations run in tf_sdf_forward."""
    dev = field.device
    gx, gy, gz = grid
    lin = [torch.linspace(0, 1, g, device=dev) for g in grid]
    samples = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)
    lo, hi = field.aabb_dev[0], field.aabb_dev[1]
    xyz = lo * (1 - samples) + hi * samples                                    # [gx,gy,gz,3]
    length = float(((field.aabb[1] - field.aabb[0]) / (torch.tensor(grid, dtype=torch.float32) - 1)).mean())
    flat = xyz.reshape(-1, 3)
    alpha = torch.zeros(flat.shape[0], device=dev)
    for c0 in range(0, flat.shape[0], chunk):
        p = flat[c0:c0 + chunk].contiguous()
        live = prev.alive(p) if prev is not None else torch.ones(p.shape[0], dtype=torch.bool, device=dev)
        sdf = field.sdf(p, None)
        pc = torch.sigmoid((sdf + length * 0.5) * inv_s)
        nc = torch.sigmoid((sdf - length * 0.5) * inv_s)
        a = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
        a = torch.where(sdf.abs() < mul_length * length, torch.ones_like(a), a)
        alpha[c0:c0 + chunk] = torch.where(live, a, torch.zeros_like(a))
    raw = alpha.view(gx, gy, gz)                                                                    # compute_gridAlpha's lattice
    alpha = raw.clamp(0, 1).transpose(0, 2).contiguous()[None, None]                               # [1,1,gz,gy,gx]
    alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1)[0, 0]
    vol = alpha >= thres
    gxyz = xyz.transpose(0, 2)
    valid = gxyz[vol]
    new_aabb = torch.stack([valid.amin(0), valid.amax(0)]) if valid.numel() else field.aabb_dev.clone()
    if return_alpha:
        return AlphaMask(field.aabb, vol), new_aabb, raw
    return AlphaMask(field.aabb, vol), new_aabb


class OccGrid(torch.nn.Module):
    """Occupancy grid of the `use_occ_grid` shape configs (configs/shape/syn/compressor_occ.yaml:21): the role nerfacc.OccGridEstimator
    plays in the reference (shapeRenderer.py:213-216 construction, :950-959 sampling, :1286-1290 update_every_n_steps, :343-353
    checkpoint).  nerfacc is third-party and absent (parity unpinned); this class follows its published semantics
    -- one level, `occs` = EMA of the evaluated opacity per cell, `binaries` = occs > min(mean(occs), occ_thre) -- with the state on
    the device and the sampling done by tf_march_uniform (cell lookup, stratified start), pinned by the build's own oracle
    (oracle/march.py:occ_grid_update, march_uniform(cells=True))."""

    def __init__(self, aabb, resolution=128, device="cuda"):
        super().__init__()
        res = [int(resolution)] * 3 if isinstance(resolution, int) else [int(r) for r in resolution]
        self.device = device
        self.n_cells = res[0] * res[1] * res[2]
        # the six registered buffers of nerfacc.OccGridEstimator, by name, dtype and shape: the reference keeps the estimator as a
        # sub-module of ShapeRenderer, so its state_dict() carries `occ_grid.<buffer>` keys and the strict load expects them
        self.register_buffer("resolution", torch.tensor(res, dtype=torch.int32, device=device))
        self.register_buffer("aabbs", torch.as_tensor(aabb, dtype=torch.float32).reshape(1, 6).to(device))
        self.register_buffer("occs", torch.zeros(self.n_cells, device=device))
        self.register_buffer("binaries", torch.zeros(1, *res, dtype=torch.bool, device=device))
        g = torch.stack(torch.meshgrid(*[torch.arange(r, device=device) for r in res], indexing="ij"), -1).reshape(-1, 3)
        self.register_buffer("grid_coords", g)
        self.register_buffer("grid_indices", torch.arange(self.n_cells, device=device))
        self.gen = None                      # torch.Generator on the device (tests seed it); None = global RNG

    # ---- nerfacc.OccGridEstimator._update / update_every_n_steps
    def _cells_to_update(self, step, warmup_steps):
        if step < warmup_steps:
            return self.grid_indices
        n = self.n_cells // 4
        uniform = torch.randint(self.n_cells, (n,), device=self.device, generator=self.gen)
        occupied = torch.nonzero(self.binaries.reshape(-1))[:, 0]
        if occupied.numel() > n:
            occupied = occupied[torch.randint(occupied.numel(), (n,), device=self.device, generator=self.gen)]
        return torch.cat([uniform, occupied])

    @torch.no_grad()
    def update_every_n_steps(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        if not self.training or step % n != 0:
            return False
        idx = self._cells_to_update(step, warmup_steps)
        coords = self.grid_coords[idx].float()
        u = (coords + torch.rand(coords.shape, device=self.device, generator=self.gen)) / self.resolution.float()
        lo, hi = self.aabbs[0, :3], self.aabbs[0, 3:]
        x = lo + u * (hi - lo)
        occ = occ_eval_fn(x).reshape(-1)
        self.occs[idx] = torch.maximum(self.occs[idx] * ema_decay, occ)
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        self.binaries = (self.occs > thre).view(self.binaries.shape)
        return True

    # ---- nerfacc.OccGridEstimator.sampling -> (ray_indices, t_starts, t_ends)
    @torch.no_grad()
    def sampling(self, rays_o, rays_d, near_plane=0.0, far_plane=1e10, render_step_size=1e-3, stratified=False, max_steps=1024):
        rn = rays_o.shape[0]
        near = torch.full((rn,), float(near_plane), device=rays_o.device)
        far = torch.full((rn,), float(far_plane), device=rays_o.device)
        jit = torch.rand(rn, device=rays_o.device, generator=self.gen) * render_step_size if stratified else None
        aabb = self.aabbs[0].reshape(2, 3).cpu()
        t0, t1, ridx = ops.march_uniform(rays_o, rays_d, near, far, aabb, max_steps, render_step_size,
                                         self.binaries[0].to(torch.uint8).contiguous(), aabb, cells=True, t_jitter=jit)
        return ridx, t0, t1

    # ---- checkpoint: nn.Module.state_dict() / load_state_dict() over the registered buffers (the `occ_grid_state_dict` entry of the
    # reference's checkpoint, shapeRenderer.py:349-353, and the `occ_grid.*` keys inside its network_state_dict)
    def load_state_dict(self, sd, strict=True):
        if "resolution" in sd and tuple(int(v) for v in sd["resolution"]) != tuple(int(v) for v in self.resolution):
            raise RuntimeError(f"OccGrid: checkpoint resolution {sd['resolution'].tolist()} != {self.resolution.tolist()}")
        return super().load_state_dict(sd, strict=strict)


@torch.no_grad()
def march_uniform(field: SdfField, o, d, near, far, n_steps=256, step_size=0.0, mask: AlphaMask = None):
    """Fixed-step sampler with occupancy culling and per-wavefront compaction (tf_march_uniform) -> packed samples."""
    return ops.march_uniform(o, d, near, far, field.aabb, n_steps, step_size, None if mask is None else mask.volume,
                             None if mask is None else mask.aabb)


@torch.no_grad()
def render_core(field: SdfField, o, d, radiis, rays_cos, t0, t1, ridx, base_radii, inv_s, cos_anneal, shade_fn=None,
                mask: AlphaMask = None, is_train=True, precision=None):
    """Forward of ShapeRenderer.render_core with a white background.  is_train=True also returns the hessian regulariser
    term (TensoSDF.gradient computes it only when training, fields.py:244-258); is_train=False (nvs / eval) skips it.
    precision: decoder arithmetic (ops.PREC_F16X3 default, ops.PREC_F32 exact).
    shade_fn(points, normals, view_dirs, feat) -> color [N,3] (split-sum shading); None -> white (geometry-only march).
    mask: AlphaGridMask culling of the packed samples (shapeRenderer.py:1119-1129)."""
    rn = o.shape[0]
    if mask is not None:
        keep = mask.alive((o[ridx] + d[ridx] * ((t0 + t1) * 0.5)[:, None]).contiguous())
        t0, t1, ridx = t0[keep], t1[keep], ridx[keep]
    mid = (t0 + t1) * 0.5
    dists = t1 - t0
    ro, rd = o[ridx], d[ridx]
    pts = ro + rd * mid[:, None]
    lv = torch.log2(ball_radii(mid[:, None], radiis[ridx], rays_cos[ridx]) / base_radii)[:, 0]
    alpha, grad, feat, sdf, nh = field.sdf_alpha(pts, lv, dists, rd, inv_s, cos_anneal, want_hess=is_train, precision=precision)
    color = shade_fn(pts, F.normalize(grad, dim=-1), -rd, feat) if shade_fn is not None else torch.ones_like(pts)
    vals = torch.cat([color, grad], -1).contiguous()
    w, acc, out = ops.composite(alpha, ridx, vals, rn)
    acc = acc[:, None]
    rgb = out[:, :3] + (1 - acc)
    nrm = F.normalize(out[:, 3:6] * acc + (1.0 - acc) * torch.tensor([0.0, 0.0, 1.0], device=o.device), dim=-1)
    return dict(ray_rgb=rgb, acc=acc, normal=nrm, gradient_error=(grad.norm(dim=-1) - 1.0) ** 2, alpha=alpha, weights=w,
                sdf=sdf, grad=grad, feat=feat, normal_hessian=nh, points=pts, levels=lv,
                loss_sparse=torch.exp(-20.0 * sdf.abs()).mean(), loss_hessian=None if nh is None else nh.abs().mean())
