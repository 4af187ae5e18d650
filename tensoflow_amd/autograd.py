"""torch.autograd.Function wrappers: forward AND backward run in libtensoflow_hip.so
(one Function per fused kernel group, inputs saved and recomputed in backward, like the reference's
renderutils ops, network/renderutils/ops.py:391-402)."""
import torch

from . import ops


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


class FlowLogqFn(torch.autograd.Function):
    """(z, logq) = TensoFlow.forward given cond; backward = tf_flow_logq_bwd (fused HIP reverse pass)."""

    @staticmethod
    def forward(ctx, cond, x, rays_id, *wb):
        weights = [[(wb[8 * k + 2 * l], wb[8 * k + 2 * l + 1]) for l in range(4)] for k in range(2)]
        z, logq = ops.flow_logq(weights, cond.detach(), x, rays_id=rays_id, precision=ops.PREC_F32)
        ctx.save_for_backward(cond, x, *wb)
        ctx.rays_id = rays_id
        ctx.mark_non_differentiable(z)
        return z, logq

    @staticmethod
    def backward(ctx, g_z, g_logq):
        cond, x, *wb = ctx.saved_tensors
        weights = [[(wb[8 * k + 2 * l], wb[8 * k + 2 * l + 1]) for l in range(4)] for k in range(2)]
        grads, g_cond = ops.flow_logq_bwd(weights, cond.detach(), x, g_logq.contiguous(), rays_id=ctx.rays_id)
        flat = []
        for k in range(2):
            for l in range(4):
                flat += [grads[k][l][0], grads[k][l][1]]
        return (g_cond, None, None, *flat)


class ShadeWeightsFn(torch.autograd.Function):
    """wgt [pn,T,3] = BRDF weight / pdf / count per direction slot (tf_shade_dirs); differentiable wrt the per-point materials
    (tf_shade_dirs_bwd).  Directions, masks, the live flags and the NIS log-Jacobian ride along as non-differentiable outputs."""

    @staticmethod
    def forward(ctx, metallic, roughness, albedo, normals, view, ang_d, logq_d, fixed_d, ang_s, logq_s, az_jitter):
        dirs, wgt, smask, live, logjac = ops.shade_dirs(normals, view, metallic.detach(), roughness.detach(), albedo.detach(), ang_d,
                                                        logq_d, fixed_d, ang_s, logq_s, az_jitter=az_jitter, want_logjac=True)
        ctx.save_for_backward(metallic, roughness, albedo, normals, view, dirs, wgt)
        ctx.sizes = (ang_d.shape[1], fixed_d.shape[0], ang_s.shape[1])
        ctx.mark_non_differentiable(dirs, smask, live, logjac)
        return wgt, dirs, smask, live, logjac

    @staticmethod
    def backward(ctx, g_wgt, *unused):
        metallic, roughness, albedo, normals, view, dirs, wgt = ctx.saved_tensors
        sd, nf, ss = ctx.sizes
        g_alb, g_met, g_rough = ops.shade_dirs_bwd(normals, view, metallic, roughness, albedo, dirs, wgt, g_wgt.contiguous(), sd, nf, ss)
        return (g_met.view_as(metallic), g_rough.view_as(roughness), g_alb, None, None, None, None, None, None, None, None)


class LightsFn(torch.autograd.Function):
    """lights [M,3] of MCShadingNetwork.get_lights (fields.py:951-975): BVH visibility, cube-map light on a miss, inner-light MLP
    on a hit, near mask -- forward entirely in the HIP kernels.  Backward: the cube-map gradient is tf_cube_lookup_bwd (scatter);
    the inner-light weight gradients are plain library GEMMs on the [hits,123] encoding produced by tf_inner_light_encode."""

    @staticmethod
    def forward(ctx, env_base, pts_rep, dirs, live, bvh, unit, exp_max, precision, *inner_wb):
        inters, nrm, depth, hit = bvh.trace(pts_rep, dirs, 1e-5, 2 * unit, live=live)
        lights = ops.cube_lookup(env_base.detach(), dirs, apply_exp=True, depth=depth, near_eps=1e-5)
        idx, count = ops.compact_mask(hit.view(torch.uint8))
        weights = [(inner_wb[2 * l].detach(), inner_wb[2 * l + 1].detach()) for l in range(4)]
        ops.inner_light_indexed(weights, inters, dirs, nrm, idx, count, depth, lights, near_eps=1e-5, exp_max=exp_max, precision=precision)
        ctx.save_for_backward(env_base, dirs, inters, nrm, depth, hit, idx, count, *inner_wb)
        ctx.exp_max = exp_max
        ctx.mark_non_differentiable(hit)
        return lights, hit

    @staticmethod
    def backward(ctx, g_lights, _g_hit):
        env_base, dirs, inters, nrm, depth, hit, idx, count, *inner_wb = ctx.saved_tensors
        near = (depth > 1e-5).float()[:, None]
        g = (g_lights * near).contiguous()
        g_base = ops.cube_lookup_bwd(env_base.detach(), dirs, g * (~hit).float()[:, None], apply_exp=True)
        grads = [None] * 8
        n_hit = int(count.item())                                    # training only: one sync per step
        if n_hit > 0:
            rows = idx[:n_hit]
            X = ops.inner_light_encode(inters, dirs, nrm, idx, count)[:n_hit]
            with torch.enable_grad():
                wb = [t.detach().requires_grad_(True) for t in inner_wb]
                h = X
                for l in range(4):
                    h = torch.nn.functional.linear(h, wb[2 * l], wb[2 * l + 1])
                    if l < 3:
                        h = torch.relu(h)
                out = torch.exp(h.clamp(max=ctx.exp_max))
                grads = list(torch.autograd.grad(out, wb, g[rows]))
        return (g_base, None, None, None, None, None, None, None, *grads)
