"""TensoFlow (reference: network/flow.py:643-855) on the MI355X kernels."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..autograd import FlowLogqFn, VmGatherFn
from ..shading import posenc, sphere_latent, sphere_latent_on


class Reshift(nn.Module):
    """network/flow.py:146-164 (kept only so that `flows.k.nn.0.{scale,offset}` exist in the state_dict)."""

    def __init__(self, scale=2.0, offset=-1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.scalar_tensor(scale), requires_grad=False)
        self.offset = nn.Parameter(torch.scalar_tensor(offset), requires_grad=False)

    def forward(self, x):
        return x * self.scale + self.offset


class Block(nn.Module):
    """Coupling block container (network/flow.py:549-598): nn = [Reshift (not for 'realnvp'), Linear, LeakyReLU, ... , Linear]."""

    def __init__(self, d, mask, feature_dim, multires=3, d_hidden=64, n_hidden=3, n_bins=21, reshift=True):
        super().__init__()
        self.d, self.mask = d, mask
        d_in = sum(mask) * (1 + 2 * multires)
        layers = [Reshift()] if reshift else []          # (flow.py:646: 'realnvp' has no input activation -- its Linear layers sit at 0, 2, 4, 6)
        self.first = len(layers)
        last = d_in + feature_dim
        for _ in range(n_hidden):
            layers += [nn.Linear(last, d_hidden), nn.LeakyReLU()]
            last = d_hidden
        layers.append(nn.Linear(last, (d - sum(mask)) * n_bins))
        self.nn = nn.Sequential(*layers)


def _check_no_grad(module, what):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise RuntimeError(f"{what}: this method's fused HIP forward has no backward (the differentiable routes are autograd.SdfAlphaFn "
                           "for the SDF field inside the renderers and TensoFlow.forward / autograd.FlowLogqFn for the flows' log-density) "
                           "-- call it under torch.no_grad() or freeze the module (the reference samples from frozen `*_copy` flows only, "
                           "fields.py:1054-1065)")


class TensoFlow(nn.Module):
    def __init__(self, d, aabb, device="cuda", gridSize=[512, 512, 512], nis_n_comp=12, nis_dim=64, nis_feature_dim=16,
                 nis_multires=3, refl_multires=3, roughness_multires=3, angle_multires=3, flow="pwquad", n_bins=10,
                 disable_tensorial=False, disable_reflected=False):
        super().__init__()
        from .flow_transforms import TRANSFORMS
        if flow not in TRANSFORMS or d != 2 or int(n_bins) < 2 or nis_dim != 64 or nis_feature_dim != 16 or \
                (nis_multires, refl_multires, roughness_multires, angle_multires) != (3, 3, 3, 3):
            raise NotImplementedError("TensoFlow: d=2, flow in ('pwquad', 'pwlinear', 'realnvp'), n_bins >= 2, 64/16 dims, 3 octaves per embedding")
        # The fused HIP kernels instantiate the reference default ('pwquad', 10 bins: what every shipped config runs).  Any other
        # (flow, n_bins) is evaluated by a differentiable COMPOSITION (flow_transforms.py: torch ops on the device around the HIP VM gather
        # and dense layers) -- correct, not fast; round 6.  'realnvp' (flow.py:645): Gaussian latent prior, affine couplings without the
        # Reshift input activation, and the analytic-sigmoid output cell behind the two blocks.
        self.flow_kind, self.n_bins = flow, int(n_bins)
        self._fused = flow == "pwquad" and int(n_bins) == 10
        self._sample_fn, self._density_fn, bin_fn = TRANSFORMS[flow]
        self.nis_n_comp, self.nis_dim, self.nis_feature_dim = nis_n_comp, nis_dim, nis_feature_dim
        self.device = device
        self.matMode, self.vecMode = [[0, 1], [0, 2], [1, 2]], [2, 1, 0]
        self.gridSize = torch.tensor(gridSize)
        self.aabb = aabb
        self.n_levels = 3
        planes, lines = [], []
        for i in range(3):
            ps = self.gridSize[self.matMode[i]]
            ls = self.gridSize[self.vecMode[i]]
            planes.append(nn.Parameter(1e-4 * (2 * torch.rand(1, nis_n_comp, int(ps[0]), int(ps[1])) - 1)))
            lines.append(nn.Parameter(torch.ones(1, nis_n_comp, int(ls), 1) * (1.0 / (nis_n_comp * 3))))
        self.nis_plane = nn.ParameterList(planes).to(device)
        self.nis_line = nn.ParameterList(lines).to(device)
        self.nis_mat = nn.Sequential(nn.Linear(3 * nis_n_comp + 21, nis_dim), nn.Softplus(beta=100),
                                     nn.Linear(nis_dim, nis_feature_dim)).to(device)
        self.refl_input_ch, self.roughness_input_ch = 14, 7
        feature_dim = nis_feature_dim + self.refl_input_ch + self.roughness_input_ch
        self.gaussian = flow == "realnvp"
        self.flows = nn.ModuleList([Block(d, [(i + off) % 2 == 0 for i in range(d)], feature_dim, n_bins=bin_fn(int(n_bins)),
                                          reshift=not self.gaussian) for off in range(2)]).to(device)
        self.disable_tensorial, self.disable_reflected = disable_tensorial, disable_reflected
        self._packed = None
        self._packed_version = None

    # ---- parameter plumbing
    def _field(self):
        ver = tuple(p._version for p in list(self.nis_plane) + list(self.nis_line))
        if self._packed is None or ver != self._packed_version:
            self._packed = ops.VmPacked(list(self.nis_plane), list(self.nis_line), self.n_levels)
            self._packed_version = ver
        return self._packed

    def _nets(self):
        return [[(blk.nn[l].weight, blk.nn[l].bias) for l in (1, 3, 5, 7)] for blk in self.flows]

    def get_optparam_groups(self, lr_init_spatialxyz=0.01, lr_init_network=0.001):
        return [{"params": self.nis_line, "lr": lr_init_spatialxyz}, {"params": self.nis_plane, "lr": lr_init_spatialxyz},
                {"params": self.nis_mat.parameters(), "lr": lr_init_network}, {"params": self.flows.parameters(), "lr": lr_init_network}]

    def tenso_feature(self, xyz_sampled, level_vol=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in list(self.nis_plane) + list(self.nis_line)):
            feat = VmGatherFn.apply(xyz_sampled.reshape(-1, 3).contiguous(), level_vol, self.aabb, self.n_levels,
                                    *self.nis_plane, *self.nis_line)
        else:
            feat = ops.vm_gather(self._field(), xyz_sampled.reshape(-1, 3), level_vol, self.aabb)
        h = torch.cat([feat, posenc(xyz_sampled, 3)], -1)
        if h.is_cuda:                                   # 57-64-16 feature net on the HIP dense-layer kernels (fwd + bwd)
            from ..autograd import mlp_apply
            return mlp_apply(self.nis_mat, h.contiguous())
        return self.nis_mat(h)

    def _condition(self, pts, reflections):
        feature = self.tenso_feature(pts)
        if self.disable_tensorial:
            feature = torch.zeros_like(feature)
        refl = posenc(reflections, 3)
        if self.disable_reflected:
            refl = torch.zeros_like(refl)
        return torch.cat([feature, refl, torch.zeros(pts.shape[0], 7, device=pts.device)], -1).contiguous()

    # ---- the composition that serves every (flow, n_bins) the fused kernels do not instantiate
    def _coupling_net(self, blk, keep, cond):
        """Block.nn on [embed3(kept coordinate), condition row] (flow.py:600-609): Reshift, then Linear + LeakyReLU x 3, Linear."""
        from ..autograd import LinearActFn
        h = torch.cat([posenc(keep, 3), cond], -1)
        f0 = blk.first
        h = (blk.nn[0](h) if f0 else h).contiguous()
        for l in (f0, f0 + 2, f0 + 4):
            h = torch.nn.functional.leaky_relu(LinearActFn.apply(h, blk.nn[l].weight, blk.nn[l].bias, ops.ACT_NONE, 0.0, None), 0.01)
        return LinearActFn.apply(h.contiguous(), blk.nn[f0 + 6].weight, blk.nn[f0 + 6].bias, ops.ACT_NONE, 0.0, None)

    def _composed_blocks(self, y, logj, cond, sampling):
        """TensoFlow.flow / flow_inv (flow.py:766-799) over rows y [M,2], cond [M,37]: the blocks in order with the sampling transform, in
        reverse order with the density transform.  -> (y, logj, bins [M,2] int64: column b = the bin block b picked)."""
        from .flow_transforms import sigmoid_cell_density, sigmoid_cell_sample
        bins = torch.zeros(y.shape[0], 2, dtype=torch.long, device=y.device)
        order = list(enumerate(self.flows)) if sampling else list(enumerate(self.flows))[::-1]
        if self.gaussian and not sampling:          # the output cell is the LAST flow (flow.py:676-677): first on the way back
            y, lj = sigmoid_cell_density(y)
            logj = logj + lj
        for bi, blk in order:
            mask = torch.tensor(blk.mask, device=y.device)
            keep, move = y[:, mask], y[:, ~mask]
            st = self._coupling_net(blk, keep, cond).view(y.shape[0], move.shape[1], -1)
            new, lj, idx = (self._sample_fn if sampling else self._density_fn)(move, st)
            out = torch.zeros_like(y)
            out[:, mask] = keep
            out[:, ~mask] = new
            y, logj = out, logj + lj
            bins[:, bi] = idx[:, 0]
        if self.gaussian and sampling:
            y, lj = sigmoid_cell_sample(y)
            logj = logj + lj
        return y, logj, bins

    def _composed_sample(self, pts, view_angles, n_samples, jitter=None):
        """sample() as a composition: SphereSampler (flow.py:52-90) -> the blocks with the sampling transform."""
        pn = pts.shape[0]
        if self.gaussian:            # FactorizedGaussianSampler.forward (flow.py:21-24): fresh normal draws, logj = -log_prob
            from .flow_transforms import gaussian_log_prob
            x = self._gaussian_latent(pn, n_samples, pts.device)
            logj = -gaussian_log_prob(x)
        else:
            x = sphere_latent_on(n_samples, pts.device)[None].expand(pn, n_samples, 2)
            if jitter is not None:
                x = torch.cat([(x[..., :1] + jitter[..., None]) % 1, x[..., 1:]], -1)
            x = x.clamp(1e-6, 1 - 1e-6)
            logj = -torch.cos(x[..., 1:] * (0.5 * np.pi)).log()
        cond = self._condition(pts, view_angles)[:, None].expand(pn, n_samples, 37).reshape(-1, 37)
        y, lj, bins = self._composed_blocks(x.reshape(-1, 2), logj.reshape(-1, 1), cond, sampling=True)
        self.last_bins = bins.view(pn, n_samples, 2)
        return y.view(pn, n_samples, 2), lj.view(pn, n_samples, 1)

    def _gaussian_latent(self, pn, n_samples, device):
        """The latent draws of the 'realnvp' prior [pn, n_samples, 2] (one place, so that a test can substitute recorded draws)."""
        return torch.randn(pn, n_samples, 2, device=device)

    def _composed_forward(self, pts, reflections, x, rays_id):
        """forward() as a composition (flow.py:801-831): the blocks in reverse with the density transform, + log prior of z."""
        cond = self._condition(pts, reflections)
        shape = x.shape[:-1]
        if rays_id is not None:
            cond_rows = cond[rays_id]
        else:
            cond_rows = cond[:, None].expand(*shape, 37).reshape(-1, 37) if x.dim() == 3 else cond
        xr = x.clamp(1e-6, 1 - 1e-6).reshape(-1, 2)
        z, lj, bins = self._composed_blocks(xr, torch.zeros(xr.shape[0], 1, device=xr.device), cond_rows, sampling=False)
        self.last_bins = bins.view(*shape, 2)
        if self.gaussian:
            from .flow_transforms import gaussian_log_prob
            logq = lj + gaussian_log_prob(z)
        else:
            logq = lj + torch.cos(z[:, 1:] * (0.5 * np.pi)).log()
        return z.view(*shape, 2), logq.view(*shape, 1)

    @torch.no_grad()
    def _sample_nograd(self, pts, view_angles, n_samples, jitter=None):
        """sample() for a frozen copy (fields.py:1054-1065): angles [pn,sn,2], logq [pn,sn,1]."""
        if not self._fused:
            return self._composed_sample(pts, view_angles, n_samples, jitter)
        cond = self._condition(pts, view_angles)
        return ops.flow_sample(self._nets(), cond, sphere_latent_on(n_samples, pts.device), jitter, precision=ops.PREC_F16X3)

    # ---- reference API
    def sample(self, pts, reflections, roughness, n_samples, return_jacobian=False):
        """flow.py:833-855 -> angles [pn,sn,2] (, logj [pn,sn,1])."""
        jitter = torch.rand(pts.shape[0], n_samples, device=pts.device) if self.training else None   # flow.py:86-87
        if not self._fused:
            ang, logj = self._composed_sample(pts, reflections, n_samples, jitter)
            return (ang, logj) if return_jacobian else ang
        _check_no_grad(self, "TensoFlow.sample")
        ang, logj = ops.flow_sample(self._nets(), self._condition(pts, reflections), sphere_latent_on(n_samples, pts.device), jitter)
        return (ang, logj) if return_jacobian else ang

    def forward(self, pts, reflections, roughness, x, return_jacobian=False, rays_id=None):
        """flow.py:801-831 -> z (, logqx).  Differentiable wrt every parameter of the flow (the NIS loss path):
        forward and backward are fused HIP kernels (autograd.FlowLogqFn / VmGatherFn); the 57-64-16 feature net runs on the
        HIP dense-layer kernels (autograd.mlp_apply)."""
        if not self._fused:
            z, logq = self._composed_forward(pts, reflections, x, rays_id)
            return (z, logq) if return_jacobian else z
        cond = self._condition(pts, reflections)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            wb = []
            for blk in self.flows:
                for l in (1, 3, 5, 7):
                    wb += [blk.nn[l].weight, blk.nn[l].bias]
            z, logq = FlowLogqFn.apply(cond, x.clamp(1e-6, 1 - 1e-6).contiguous(), rays_id, *wb)
        else:
            z, logq = ops.flow_logq(self._nets(), cond, x, rays_id=rays_id)
        return (z, logq) if return_jacobian else z
