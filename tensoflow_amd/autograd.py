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
