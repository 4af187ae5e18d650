"""torch.autograd.Function wrappers: forward AND backward run in libtensoflow_hip.so
(one Function per fused kernel group, inputs saved and recomputed in backward, like the reference's
renderutils ops, network/renderutils/ops.py:391-402)."""
import torch

from . import ops


class LinearActFn(torch.autograd.Function):
    """y = act(x w^T + b): forward tf_linear_fwd, backward tf_linear_bwd (data, weight and bias gradients) -- the dense layers of
    every MLP in a training step run on the exact-fp32 matrix cores of libtensoflow_hip.so, none on a library GEMM.
    n_dev: optional device-side row count (compacted hit lists): rows beyond it are neither computed nor differentiated."""

    @staticmethod
    def forward(ctx, x, w, b, act, act_param, n_dev=None):
        x, w = x.contiguous(), w.contiguous()
        y = ops.linear_fwd(x.detach(), w.detach(), None if b is None else b.detach(), act, act_param, n_dev=n_dev)
        ctx.save_for_backward(x, w, y)
        ctx.cfg = (act, act_param, b is not None, n_dev)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        act, act_param, has_b, n_dev = ctx.cfg
        gx, gw, gb = ops.linear_bwd(x, w, y, gy.contiguous(), act, act_param, need_gx=ctx.needs_input_grad[0],
                                    need_gw=ctx.needs_input_grad[1], need_gb=has_b and ctx.needs_input_grad[2], n_dev=n_dev)
        return gx, gw, gb, None, None, None


class MlpChainFn(torch.autograd.Function):
    """A whole stack of Linear + activation layers as ONE autograd node: forward = tf_linear_fwd per layer (outputs kept), backward =
    tf_linear_bwd_fused down the stack -- each layer's data-gradient product also runs the activation backward of the layer below, so
    no pass over the [rows, width] gradients remains between the layers (LinearActFn per layer: one such pass per layer).
    args: x, n_dev, acts (tuple of (act, param) per layer), then weight, bias (or None) per layer."""

    @staticmethod
    def forward(ctx, x, n_dev, acts, *wb):
        hs = [x.contiguous()]
        ws = []
        for l, (act, prm) in enumerate(acts):
            w, b = wb[2 * l].contiguous(), wb[2 * l + 1]
            ws.append(w)
            hs.append(ops.linear_fwd(hs[-1].detach(), w.detach(), None if b is None else b.detach(), act, prm, n_dev=n_dev))
        ctx.save_for_backward(*hs, *ws)
        ctx.cfg = (acts, n_dev, [b is not None for b in wb[1::2]])
        return hs[-1]

    @staticmethod
    def backward(ctx, gy):
        acts, n_dev, has_b = ctx.cfg
        L_ = len(acts)
        saved = ctx.saved_tensors
        hs, ws = saved[:L_ + 1], saved[L_ + 1:]
        grads = [None] * (2 * L_)
        g, is_gz = gy.contiguous(), False
        # the weight / bias gradients of the whole stack are accumulated (atomics) into ONE buffer zeroed with one launch
        # (every piece starts on a 256-byte boundary: the aligned dense-product kernels ask for 16-byte aligned operands)
        up = lambda k: (k + 63) // 64 * 64
        total = sum(up(w.numel()) + 2 * up(max(w.shape)) for w in ws)
        pool = torch.zeros(total, dtype=torch.float32, device=gy.device)
        used = [0]

        def take(k):
            out = pool[used[0]:used[0] + k]
            used[0] += up(k)
            assert used[0] <= total
            return out
        for l in range(L_ - 1, -1, -1):
            below = l > 0
            need_gx = below or ctx.needs_input_grad[0]
            gx, gw, gb, gbx = ops.linear_bwd_fused(hs[l], ws[l], hs[l + 1], g, acts[l][0], acts[l][1], gy_is_gz=is_gz,
                                                   x_act=acts[l - 1][0] if below else ops.ACT_NONE,
                                                   x_act_param=acts[l - 1][1] if below else 0.0, need_gx=need_gx,
                                                   need_gw=ctx.needs_input_grad[3 + 2 * l], need_gb=has_b[l] and ctx.needs_input_grad[4 + 2 * l],
                                                   need_gbx=below and has_b[l - 1] and ctx.needs_input_grad[4 + 2 * (l - 1)], n_dev=n_dev,
                                                   zeroed=take)
            grads[2 * l] = gw
            if not is_gz:
                grads[2 * l + 1] = gb
            if below:
                grads[2 * (l - 1) + 1] = gbx
            g, is_gz = gx, below
        return (g if ctx.needs_input_grad[0] else None, None, None, *grads)


def mlp_apply(seq, x, n_dev=None):
    """Evaluate an nn.Sequential of Linear / activation modules (make_predictor_3layer / _4layer and TensoSDF.sdf_mat shapes) with
    every Linear + its following activation as ONE LinearActFn.  The modules keep their parameters (weight-norm parametrizations
    included: `layer.weight` composes g v / |v| under autograd) and their state_dict keys; only the arithmetic moves."""
    mods = list(seq)
    i = 0
    layers = []
    while i < len(mods):
        m = mods[i]
        if not isinstance(m, torch.nn.Linear):
            raise NotImplementedError(f"mlp_apply: unexpected module {type(m).__name__} at position {i}")
        act, prm, step = ops.ACT_NONE, 0.0, 1
        if i + 1 < len(mods) and not isinstance(mods[i + 1], torch.nn.Linear):
            a = mods[i + 1]
            step = 2
            if isinstance(a, torch.nn.ReLU):
                act = ops.ACT_RELU
            elif isinstance(a, torch.nn.Softplus):
                act, prm = ops.ACT_SOFTPLUS, float(a.beta)
            elif isinstance(a, torch.nn.Sigmoid):
                act = ops.ACT_SIGMOID
            elif isinstance(a, torch.nn.Identity):
                act = ops.ACT_NONE
            elif hasattr(a, "max_light"):                       # ExpActivation (other_field.py:12-18)
                act, prm = ops.ACT_EXP_CLAMP, float(a.max_light)
            else:
                raise NotImplementedError(f"mlp_apply: activation {type(a).__name__}")
        layers.append((m, act, prm))
        i += step
    if len(layers) == 1:
        m, act, prm = layers[0]
        return LinearActFn.apply(x, m.weight, m.bias, act, prm, n_dev)
    wb = []
    for m, _, _ in layers:
        wb += [m.weight, m.bias]
    return MlpChainFn.apply(x, n_dev, tuple((a, p) for _, a, p in layers), *wb)


class VmGatherFn(torch.autograd.Function):
    """feat = gather(planes, lines)(xyz, level): forward tf_vm_gather_fwd on the packed pyramid; backward
    tf_vm_gather_bwd (float atomics into a pyramid-shaped buffer) + tf_vm_pack_bwd (box-filter adjoint, layout restore)."""

    @staticmethod
    def forward(ctx, xyz, level, aabb, n_levels, *fields):
        planes, lines = list(fields[:3]), list(fields[3:6])
        packed = ops.VmPacked(planes, lines, n_levels)
        ctx.packed, ctx.aabb = packed, aabb
        ctx.save_for_backward(xyz, level if level is not None else torch.empty(0, device=xyz.device), *fields)
        ctx.has_level = level is not None
        return ops.vm_gather(packed, xyz, level, aabb)

    @staticmethod
    def backward(ctx, gfeat):
        xyz, level, *fields = ctx.saved_tensors
        gp = ops.vm_gather_bwd(ctx.packed, xyz, level if ctx.has_level else None, ctx.aabb, gfeat.contiguous())
        gplanes, glines = ctx.packed.unpack_grad(gp, fields[:3], fields[3:6])
        return (None, None, None, None, *gplanes, *glines)


class SrgbFn(torch.autograd.Function):
    """linear_to_srgb (utils/raw_utils.py:4-17), optionally followed by clamp(., 0, 1): one launch each way."""

    @staticmethod
    def forward(ctx, lin, clamp01):
        ctx.save_for_backward(lin)
        ctx.clamp01 = bool(clamp01)
        return ops.linear_to_srgb(lin, clamp01)

    @staticmethod
    def backward(ctx, g):
        (lin,) = ctx.saved_tensors
        return ops.linear_to_srgb(lin, ctx.clamp01, g_out=g.contiguous()), None


def linear_to_srgb(lin, clamp01=False):
    """The device form on a HIP tensor (one launch, analytic backward); the torch composition elsewhere (CPU tests of the host logic)."""
    if lin.is_cuda:
        return SrgbFn.apply(lin, clamp01)
    from .encodings import linear_to_srgb as composed
    y = composed(lin)
    return y.clamp(0, 1) if clamp01 else y


class Normalize3Fn(torch.autograd.Function):
    """F.normalize(x, dim=-1) of [n,3] rows -- optionally of the blend x acc + (1 - acc) c, optionally with the eikonal residual
    (|x| - 1)^2 -- as one launch each way (render_core: the samples' normals + gradient_error, the composited ray normal)."""

    @staticmethod
    def forward(ctx, x, acc, blend_c, want_err):
        ctx.set_materialize_grads(False)
        y, err = ops.normalize3(x, acc, blend_c, want_err)
        ctx.save_for_backward(x, acc)
        ctx.blend_c = blend_c
        if err is None:
            err = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(err)
        return y, err

    @staticmethod
    def backward(ctx, g_y, g_err):
        x, acc = ctx.saved_tensors
        if g_y is None and g_err is None:
            return None, None, None, None
        gx, ga = ops.normalize3(x, acc, ctx.blend_c, grads=(g_y, g_err))
        return gx, (ga.reshape(acc.shape) if acc is not None else None), None, None


class ShapeGluePreFn(torch.autograd.Function):
    """ShapeShadingNetwork.forward's prelude (fields.py:455-463) as one launch each way: unit normals (degenerate rows patched), unit
    view, NoV, reflective, roughness and -- with mip_levels = (min_roughness, max_roughness, n_levels) -- the specular-stack coordinate
    EnvLight.get_mip(roughness).clamp(0, n - 1) (13 element-wise launches of their own before).  Differentiable wrt the normals and mat
    (the view directions of a training step carry no gradient)."""

    @staticmethod
    def forward(ctx, normals, view, mat, mip_levels=None):
        ctx.set_materialize_grads(False)
        nu, vu, nov, refl, rough, mip = ops.shape_glue_pre(normals, view, mat, mip_levels=mip_levels)
        ctx.save_for_backward(normals, view, mat)
        ctx.mip_levels = mip_levels
        ctx.mark_non_differentiable(vu)
        if mip is None:
            mip = torch.empty(0, device=normals.device)
            ctx.mark_non_differentiable(mip)
        return nu, vu, nov, refl, rough, mip

    @staticmethod
    def backward(ctx, g_nu, g_vu, g_nov, g_refl, g_rough, g_mip):
        normals, view, mat = ctx.saved_tensors
        if ctx.mip_levels is None:
            g_mip = None
        gn, gm = ops.shape_glue_pre(normals, view, mat, grads=(g_nu, g_nov, g_refl, g_rough, g_mip), mip_levels=ctx.mip_levels)
        return gn, None, gm, None


class ShapeGluePostFn(torch.autograd.Function):
    """ShapeShadingNetwork.forward's split-sum combination (fields.py:460-561) as one launch each way -> (color, occ_prob)."""

    @staticmethod
    def forward(ctx, mat, nov, diffuse_light, direct_light, indirect_light, occ_raw, fg_lut):
        ctx.set_materialize_grads(False)
        color, occ_prob = ops.shape_glue_post(mat, nov, diffuse_light, direct_light, indirect_light, occ_raw, fg_lut)
        ctx.save_for_backward(mat, nov, diffuse_light, direct_light, indirect_light, occ_raw, fg_lut)
        return color, occ_prob

    @staticmethod
    def backward(ctx, g_color, g_occ_prob):
        saved = ctx.saved_tensors
        if g_color is None:
            g_color = torch.zeros(saved[0].shape[0], 3, device=saved[0].device)
        gm, gnov, gdl, gdr, gil, gocc = ops.shape_glue_post(*saved, grads=(g_color, g_occ_prob))
        return gm, gnov, gdl, gdr, gil, gocc, None


class FlowLogqFn(torch.autograd.Function):
    """(z, logq) = TensoFlow.forward given cond; backward = tf_flow_logq_bwd (fused HIP reverse pass)."""

    @staticmethod
    def forward(ctx, cond, x, rays_id, *wb):
        weights = [[(wb[8 * k + 2 * l], wb[8 * k + 2 * l + 1]) for l in range(4)] for k in range(2)]
        # split f16 operands like the eval path (fp32-grade; the exact-fp32 kernel took 2.5x as long for the same parity: round 5)
        z, logq = ops.flow_logq(weights, cond.detach(), x, rays_id=rays_id, precision=ops.PREC_F16X3)
        ctx.save_for_backward(cond, x, z, *wb)
        ctx.rays_id = rays_id
        ctx.mark_non_differentiable(z)
        return z, logq

    @staticmethod
    def backward(ctx, g_z, g_logq):
        cond, x, z, *wb = ctx.saved_tensors
        weights = [[(wb[8 * k + 2 * l], wb[8 * k + 2 * l + 1]) for l in range(4)] for k in range(2)]
        # d logq / d x is only asked for between nis_loss_iter and nis_start_iter, where the NIS loss is fitted on the fixed GGX half
        # angles and those depend on the predicted roughness (fields.py:1296-1318 with sample_specular_directions): closed form in the
        # same kernel (round 3; central differences of the forward -- round 2 -- were exact in the median and off by more than the
        # gradient itself on the 0.03 % of samples whose stencil straddled a knot of a narrow spline bin: tools/exp_flow_dx.py of the round-4 tree, git a8fd04d)
        want_gx = ctx.needs_input_grad[1]
        res = ops.flow_logq_bwd(weights, cond.detach(), x, g_logq.contiguous(), rays_id=ctx.rays_id, want_gx=want_gx, z=z)
        grads, g_cond = res[0], res[1]
        g_x = res[2] if want_gx else None
        flat = []
        for k in range(2):
            for l in range(4):
                flat += [grads[k][l][0], grads[k][l][1]]
        return (g_cond, g_x, None, *flat)


class ShadeWeightsFn(torch.autograd.Function):
    """wgt [pn,T,3] = BRDF weight / pdf / count per direction slot (tf_shade_dirs); differentiable wrt the per-point materials
    (tf_shade_dirs_bwd).  Directions, masks, the live flags and the NIS log-Jacobian ride along as non-differentiable outputs."""

    @staticmethod
    def forward(ctx, metallic, roughness, albedo, normals, view, ang_d, logq_d, fixed_d, ang_s, logq_s, az_jitter, whole=(False, False), smith=False):
        dirs, wgt, smask, live, logjac = ops.shade_dirs(normals, view, metallic.detach(), roughness.detach(), albedo.detach(), ang_d,
                                                        logq_d, fixed_d, ang_s, logq_s, az_jitter=az_jitter, want_logjac=True, whole=whole, smith=smith)
        ctx.smith = bool(smith)
        ctx.save_for_backward(metallic, roughness, albedo, normals, view, dirs, wgt)
        ctx.sizes = (ang_d.shape[1], fixed_d.shape[0], ang_s.shape[1])
        ctx.mark_non_differentiable(dirs, smask, live, logjac)
        return wgt, dirs, smask, live, logjac

    @staticmethod
    def backward(ctx, g_wgt, *unused):
        metallic, roughness, albedo, normals, view, dirs, wgt = ctx.saved_tensors
        sd, nf, ss = ctx.sizes
        g_alb, g_met, g_rough = ops.shade_dirs_bwd(normals, view, metallic, roughness, albedo, dirs, wgt, g_wgt.contiguous(), sd, nf, ss, smith=ctx.smith)
        return (g_met.view_as(metallic), g_rough.view_as(roughness), g_alb, None, None, None, None, None, None, None, None, None, None)


class LightsFn(torch.autograd.Function):
    """lights [M,3] of MCShadingNetwork.get_lights (fields.py:951-975): BVH visibility, outer light on a miss (the cube map, or -- with
    env_base None, outer_light_version='direction' -- the 72-256-256-256-3 net on the IDE of the direction), inner-light MLP on a hit,
    near mask -- forward entirely in the HIP kernels.  Backward: the cube-map gradient is tf_cube_lookup_bwd (scatter); the weight
    gradients of the nets come from tf_linear_fwd / tf_linear_bwd on the [hits,123] encoding of tf_inner_light_encode (the [misses,72]
    IDE rows of the outer net).  wb: the inner net's 4 (weight, bias) pairs, followed by the outer net's when env_base is None."""

    @staticmethod
    def forward(ctx, env_base, pts_rep, dirs, live, bvh, unit, exp_max, precision, outer_exp_max, *wb):
        inters, nrm, depth, hit = bvh.trace(pts_rep, dirs, 1e-5, 2 * unit, live=live)
        inner_wb, outer_wb = wb[:8], wb[8:]
        idx_m = count_m = None
        if env_base is not None:
            lights = ops.cube_lookup(env_base.detach(), dirs, apply_exp=True, depth=depth, near_eps=1e-5)
        else:
            lights = torch.zeros_like(dirs)
            idx_m, count_m = ops.compact_mask((~hit).view(torch.uint8))
            ow = [(outer_wb[2 * l].detach().contiguous(), outer_wb[2 * l + 1].detach().contiguous()) for l in range(4)]
            ops.outer_light_indexed(ow, dirs, idx_m, count_m, lights, exp_max=outer_exp_max)      # near mask of a miss: 1
        idx, count = ops.compact_mask(hit.view(torch.uint8))
        weights = [(inner_wb[2 * l].detach(), inner_wb[2 * l + 1].detach()) for l in range(4)]
        # the hidden layers' activations come out of the fused forward (fp32-grade operand split) when a backward pass will want them:
        # LightsFn.backward then differentiates the dense layers on them instead of recomputing three 256-wide layers first
        acts = None
        if precision == ops.PREC_F16X3 and any(ctx.needs_input_grad[9:17]):
            acts = torch.empty(3, idx.numel(), 256, device=dirs.device)
        ops.inner_light_indexed(weights, inters, dirs, nrm, idx, count, depth, lights, near_eps=1e-5, exp_max=exp_max, precision=precision, acts=acts)
        ctx.has_env = env_base is not None
        ctx.acts = acts
        ctx.save_for_backward(env_base if ctx.has_env else idx_m, dirs, inters, nrm, depth, hit, idx, count, count_m if not ctx.has_env else count, *wb)
        ctx.exp_max, ctx.outer_exp_max = exp_max, outer_exp_max
        ctx.mark_non_differentiable(hit)
        return lights, hit

    @staticmethod
    def _net_bwd(X, gsel, ws, count, exp_max, hidden=None):
        """Differentiate the four layers on X (HIP dense-layer kernels): -> 8 gradients (weight, bias per layer).  `hidden` [3, cap, 256]:
        the hidden activations the fused forward saved; None: the four layers are recomputed first.  The row count stays on the device
        (`count`): launches are sized for the capacity and clamp to it in-kernel -- no host sync."""
        acts = [ops.ACT_RELU, ops.ACT_RELU, ops.ACT_RELU, ops.ACT_EXP_CLAMP]
        hs = [X]
        for l in range(4):
            if hidden is not None and l < 3:
                hs.append(hidden[l])
                continue
            hs.append(ops.linear_fwd(hs[-1], ws[2 * l], ws[2 * l + 1], acts[l], exp_max, n_dev=count))
        # a CHAIN: every call also runs the activation backward of the layer below in its data-gradient product (mask in the epilogue,
        # bias gradient as column sums) -- no separate pass over the [rows, 256] gradients between the layers
        grads = [None] * 8
        gy, is_gz = gsel, False
        for l in (3, 2, 1, 0):
            gx, gw, gb, gbx = ops.linear_bwd_fused(hs[l], ws[2 * l], hs[l + 1], gy, acts[l], exp_max, gy_is_gz=is_gz,
                                                   x_act=acts[l - 1] if l > 0 else ops.ACT_NONE, need_gx=l > 0, need_gbx=l > 0, n_dev=count)
            grads[2 * l] = gw
            if not is_gz:
                grads[2 * l + 1] = gb
            if l > 0:
                grads[2 * (l - 1) + 1] = gbx
            gy, is_gz = gx, l > 0
        return grads

    @staticmethod
    def backward(ctx, g_lights, _g_hit):
        first, dirs, inters, nrm, depth, hit, idx, count, count_m, *wb = ctx.saved_tensors
        near = (depth > 1e-5).float()[:, None]
        g = (g_lights * near).contiguous()
        ws = [t.detach().contiguous() for t in wb]
        g_base, g_outer = None, []
        if ctx.has_env:
            g_base = ops.cube_lookup_bwd(first.detach(), dirs, g * (~hit).float()[:, None], apply_exp=True)
        else:
            from .encodings import ide5
            idx_m = first.clamp(0, g.shape[0] - 1)                                # rows >= count_m are unspecified: any valid row
            d_m = dirs.index_select(0, idx_m)
            X = ide5(d_m, torch.zeros(d_m.shape[0], 1, device=d_m.device), wide=True).contiguous()      # [cap, 72]
            g_outer = LightsFn._net_bwd(X, g.index_select(0, idx_m), ws[8:], count_m, ctx.outer_exp_max)
        # inner-light net on the hit rows
        # rows of 128 floats (5 zero columns, the first layer's weight padded to match): aligned for the dense layers' DMA kernels
        X = ops.inner_light_encode(inters, dirs, nrm, idx, count, ld=128)         # [cap, 128], rows >= count unspecified
        gsel = g.index_select(0, idx.clamp(0, g.shape[0] - 1))                    # [cap, 3]
        wi = list(ws[:8])
        wi[0] = torch.nn.functional.pad(wi[0], (0, 5))
        grads = LightsFn._net_bwd(X, gsel, wi, count, ctx.exp_max, hidden=ctx.acts)
        ctx.acts = None
        grads[0] = grads[0][:, :123].contiguous()
        return (g_base, None, None, None, None, None, None, None, None, *grads, *g_outer)


class CompositeFn(torch.autograd.Function):
    """(weights, acc, out) = packed-ray compositing (tf_composite_fwd); backward tf_composite_bwd wrt alpha and values."""

    @staticmethod
    def forward(ctx, alpha, values, ray_indices, n_rays):
        # the kernel composites up to 8 value channels per launch; wider value rows go through in column chunks
        a, v = alpha.detach(), values.detach()
        outs = []
        for c0 in range(0, v.shape[1], 8):
            w, acc, o = ops.composite(a, ray_indices, v[:, c0:c0 + 8].contiguous(), n_rays)
            outs.append(o)
        out = outs[0] if len(outs) == 1 else torch.cat(outs, -1)
        ctx.save_for_backward(alpha, values, w, ray_indices)
        ctx.n_rays = n_rays
        return w, acc, out

    @staticmethod
    def backward(ctx, g_w, g_acc, g_out):
        alpha, values, w, ridx = ctx.saved_tensors
        # gradient arriving directly on the per-sample weights is folded in as an extra value channel of ones
        if g_w is not None:          # (checked on the device: no host sync in the training step)
            torch._assert_async((g_w == 0).all(), "CompositeFn: gradients wrt the per-sample weights are not supported; use acc / out")
        a, v = alpha.detach(), values.detach()
        ga, gvs = None, []
        for c0 in range(0, v.shape[1], 8):
            # d/d alpha of acc is only counted once (first chunk); later chunks see a zero g_acc
            gacc = g_acc.contiguous() if c0 == 0 else torch.zeros_like(g_acc)
            gai, gvi = ops.composite_bwd(a, ridx, v[:, c0:c0 + 8].contiguous(), w, gacc, g_out[:, c0:c0 + 8].contiguous(), ctx.n_rays)
            ga = gai if ga is None else ga + gai
            gvs.append(gvi)
        return ga, (gvs[0] if len(gvs) == 1 else torch.cat(gvs, -1)), None, None


def sdf_alpha_composed(planes, lines, W1, b1, W2, b2, pts, level, dists, dirs, inv_s, cos_anneal, aabb, units, n_levels):
    """Differentiable restatement of tf_sdf_alpha_fwd used ONLY inside SdfAlphaFn.backward: the gather is the HIP kernel pair
    (VmGatherFn), the two decoder products are the HIP dense-layer kernels (LinearActFn) over all 7 taps, the rest is elementwise."""
    import torch.nn.functional as F
    N = pts.shape[0]
    u = torch.as_tensor(units, dtype=torch.float32, device=pts.device)
    offs = torch.zeros(7, 3, device=pts.device)
    for ax in range(3):
        offs[1 + 2 * ax, ax] = u[ax]
        offs[2 + 2 * ax, ax] = -u[ax]
    P = (pts[None] + offs[:, None]).reshape(-1, 3).contiguous()
    lv = None if level is None else level.reshape(-1).repeat(7).contiguous()
    feat = VmGatherFn.apply(P, lv, aabb, n_levels, *planes, *lines)
    m_freq = (W1.shape[1] - feat.shape[1] - 3) // 6
    if m_freq > 0:
        # sdf_multires > 0 (fields.py:66-81, :293-299): the positional encoding of the contracted (m == 3) or raw coordinates joins the
        # VM features; differentiable torch ops (the gather's coordinates are detached in the reference as well)
        from .encodings import posenc
        a = torch.as_tensor(aabb, dtype=torch.float32, device=P.device)
        E = posenc(((P - a[0]) / (a[1] - a[0])) if m_freq == 3 else P, m_freq)
        pad = (-(feat.shape[1] + E.shape[1])) % 4
        x_in = torch.cat([feat, E] + ([torch.zeros_like(P[:, :1]).expand(-1, pad)] if pad else []), -1).contiguous()
        h = LinearActFn.apply(x_in, F.pad(W1, (0, pad)), b1, ops.ACT_SOFTPLUS, 100.0)
    else:
        # 108 + 3 inputs, one zero column (and a zero weight column) added: rows of 112 floats are aligned for the dense layers' DMA kernels
        h = LinearActFn.apply(torch.cat([feat, P, torch.zeros_like(P[:, :1])], -1), F.pad(W1, (0, 1)), b1, ops.ACT_SOFTPLUS, 100.0)
    s = LinearActFn.apply(h, W2[:1].contiguous(), b2[:1].contiguous(), ops.ACT_NONE, 0.0)[:, 0].view(7, N)
    app = LinearActFn.apply(h[:N].contiguous(), W2[1:].contiguous(), b2[1:].contiguous(), ops.ACT_NONE, 0.0)
    sdf = s[0]
    grad = torch.stack([(s[1 + 2 * ax] - s[2 + 2 * ax]) / (2 * u[ax]) for ax in range(3)], -1)
    hess = torch.stack([(s[1 + 2 * ax] + s[2 + 2 * ax] - 2 * sdf) / (u[ax] ** 2) for ax in range(3)], -1)
    nh = (grad * hess).sum(-1) / ((grad ** 2).sum(-1) + 1e-5)
    true_cos = (dirs * grad).sum(-1)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-true_cos) * cos_anneal)
    pc = torch.sigmoid((sdf - iter_cos * dists * 0.5) * inv_s)
    nc = torch.sigmoid((sdf + iter_cos * dists * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
    return alpha, grad, app, sdf, nh


class SdfAlphaFn(torch.autograd.Function):
    """ShapeRenderer.compute_sdf_alpha: forward = the fused HIP kernel (tf_sdf_alpha_fwd, which also keeps the six finite-difference
    sdf values), backward = tf_sdf_alpha_bwd: ONE entry point (closed-form adjoint of alpha / cos annealing / finite differences /
    hessian term, one recompute of the hidden layer per tap, the decoder's products on the exact-fp32 matrix cores, the 7-tap
    scatter) + tf_vm_pack_bwd; gradients for planes, lines, W1, b1, W2, b2 and inv_s.  `sdf_alpha_composed` (the torch composition of
    rounds 1-3, ~160 launches) is kept as the checker of tests/test_gpu_sdf_bwd.py and behind TENSOFLOW_SDF_BWD=composed."""

    @staticmethod
    def forward(ctx, pts, level, dists, dirs, inv_s, cos_anneal, aabb, units, n_levels, *params):
        planes, lines = list(params[:3]), list(params[3:6])
        W1, b1, W2, b2 = params[6:10]
        packed = ops.VmPacked(planes, lines, n_levels)
        ctx.composed = ops.sdf_embed_freqs(packed, W1) > 0          # sdf_multires > 0: forward and backward on the composition
        inv_host = getattr(inv_s, "_tf_host", None)                 # the caller's cached read-back (ShapeRenderer._inv_s_host)
        if inv_host is None:
            inv_host = float(inv_s)
        alpha, grad, feat, sdf, nh, taps = ops.sdf_alpha(packed, W1.detach(), b1.detach(), W2.detach(), b2.detach(), pts, level, dists, dirs,
                                                         aabb, units, inv_host, cos_anneal, want_taps=True)
        ctx.save_for_backward(pts, level if level is not None else torch.empty(0, device=pts.device), dists, dirs, inv_s, sdf, taps, *params)
        ctx.cfg = (cos_anneal, aabb, units, n_levels, level is not None)
        ctx.packed = packed
        ctx.inv_s_host = inv_host                    # (the forward already read it back for the kernel argument: no second sync in backward)
        return alpha, grad, feat, sdf, nh

    @staticmethod
    def backward(ctx, g_alpha, g_grad, g_feat, g_sdf, g_nh):
        import os
        pts, level, dists, dirs, inv_s, sdf, taps, *params = ctx.saved_tensors
        cos_anneal, aabb, units, n_levels, has_level = ctx.cfg
        if ctx.composed or os.environ.get("TENSOFLOW_SDF_BWD") == "composed":         # sdf_multires > 0 (or the dev switch): the torch composition
            with torch.enable_grad():
                leaf = [p.detach().requires_grad_(True) for p in params]
                inv = inv_s.detach().requires_grad_(True)
                outs = sdf_alpha_composed(leaf[:3], leaf[3:6], leaf[6], leaf[7], leaf[8], leaf[9], pts, level if has_level else None,
                                          dists, dirs, inv, cos_anneal, aabb, units, n_levels)
                gs = [g_alpha, g_grad, g_feat, g_sdf, g_nh]
                pairs = [(o, g) for o, g in zip(outs, gs) if g is not None]
                grads = torch.autograd.grad([o for o, _ in pairs], leaf + [inv], [g for _, g in pairs], allow_unused=True)
            return (None, None, None, None, grads[-1], None, None, None, None, *grads[:-1])
        W1, b1, W2, b2 = [p.detach() for p in params[6:10]]
        gp, g_w1, g_b1, g_w2, g_b2, g_inv = ops.sdf_alpha_bwd(ctx.packed, W1, b1, W2, b2, pts, level if has_level else None, dists, dirs, aabb,
                                                              units, ctx.inv_s_host, cos_anneal, sdf, taps, g_alpha, g_grad, g_feat, g_sdf, g_nh)
        gplanes, glines = ctx.packed.unpack_grad(gp, params[:3], params[3:6])
        return (None, None, None, None, g_inv.reshape(inv_s.shape), None, None, None, None, *gplanes, *glines, g_w1, g_b1, g_w2, g_b2)


class TvLossSumFn(torch.autograd.Function):
    """sum over several [1,C,H,W] grids of TVLoss.forward (TensoSDF.TV_loss_sdf, fields.py:133-138: three planes + three lines): per grid
    tf_tv_fwd + tf_tv_finish accumulating into one device scalar, tf_tv_bwd per grid -- 13 launches forward instead of 36 (round 5)."""

    @staticmethod
    def forward(ctx, weight, *grids):
        from . import lib as L
        lib = L.load()
        dev = grids[0].device
        total = torch.zeros(1, dtype=torch.float32, device=dev)
        part = torch.empty(lib.tf_tv_partials(), dtype=torch.float32, device=dev)
        coefs = []
        for x in grids:
            b, c, h, w = x.shape
            assert b == 1 and x.is_contiguous() and x.dtype == torch.float32
            count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
            coef = (float(weight) * 2.0 / count_h if count_h else 0.0, float(weight) * 2.0 / count_w if count_w else 0.0)
            coefs.append(coef)
            L.check(lib.tf_tv_fwd(ops._p(x), c, h, w, ops._p(part), ops._stream()), "tf_tv_fwd")
            L.check(lib.tf_tv_finish(ops._p(part), coef[0], coef[1], ops._p(total), ops._stream()), "tf_tv_finish")
        ctx.coefs = coefs
        ctx.save_for_backward(*grids)
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        from . import lib as L
        lib = L.load()
        g1 = g.reshape(1).contiguous().float()
        out = []
        for x, coef in zip(ctx.saved_tensors, ctx.coefs):
            _, c, h, w = x.shape
            gx = torch.empty_like(x)
            L.check(lib.tf_tv_bwd(ops._p(x), c, h, w, ops._p(g1), coef[0], coef[1], ops._p(gx), ops._stream()), "tf_tv_bwd")
            out.append(gx)
        return (None, *out)


class TvLossFn(torch.autograd.Function):
    """TVLoss.forward (other_field.py:170-191) of one [1,C,H,W] grid: tf_tv_fwd (+ a fixed-order sum of its 1 024 partials) and
    tf_tv_bwd -- two launches forward, one backward, instead of ~12 + autograd's slice backwards on a plane-sized tensor."""

    @staticmethod
    def forward(ctx, x, weight):
        from . import lib as L
        lib = L.load()
        b, c, h, w = x.shape
        assert b == 1 and x.is_contiguous()
        part = torch.empty(lib.tf_tv_partials(), dtype=torch.float32, device=x.device)
        L.check(lib.tf_tv_fwd(ops._p(x), c, h, w, ops._p(part), ops._stream()), "tf_tv_fwd")
        s = part.view(-1, 2).sum(0)
        count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
        ctx.coef = (float(weight) * 2.0 / count_h if count_h else 0.0, float(weight) * 2.0 / count_w if count_w else 0.0)
        ctx.save_for_backward(x)
        return s[0] * ctx.coef[0] + s[1] * ctx.coef[1]

    @staticmethod
    def backward(ctx, g):
        from . import lib as L
        (x,) = ctx.saved_tensors
        _, c, h, w = x.shape
        gx = torch.empty_like(x)
        L.check(L.load().tf_tv_bwd(ops._p(x), c, h, w, ops._p(g.reshape(1).contiguous().float()), ctx.coef[0], ctx.coef[1], ops._p(gx),
                                   ops._stream()), "tf_tv_bwd")
        return gx, None
