"""Host-side composition of the flow-sampled rendering integral on the MI355X.

Mirrors MCShadingNetwork.forward -> shade_mixed (network/fields.py:1453-1473, :1075-1235) and
MCShadingNetwork.get_lights (:951-975) with every stage on the HIP kernels of libtensoflow_hip.so:
the fused per-point stage (material predictors 108-128-{1,1,3}, flow feature nets 57-64-16, condition
rows), flow sampling, direction/pdf/BRDF construction, BVH visibility, cube-map lookup, inner-light
MLP, reduction.  PyTorch only allocates the tensors.

Parameters are taken from a reference-layout state_dict (same keys as MCShadingNetwork).
"""
import math

import numpy as np
import torch

from . import ops


def posenc(x, n_freq):
    """get_embedder (utils/network_utils.py:38-50); on the device and outside a graph: one launch (encodings.posenc -> tf_posenc_fwd)."""
    from .encodings import posenc as _pe
    return _pe(x, n_freq)


def wn_weight(sd, prefix):
    k0 = prefix + ".parametrizations.weight.original0"
    if k0 in sd:
        v = sd[prefix + ".parametrizations.weight.original1"]
        return sd[k0] * v / v.norm(dim=1, keepdim=True)
    return sd[prefix + ".weight"]


def fibonacci_samples(n):
    """MCShadingNetwork fixed direction set (fields.py:734-737 via utils/base_utils.py:869-882) -> [n,2]."""
    g = (np.sqrt(5) - 1.0) / 2.0
    num_points = int(n // (1 - 0.5))
    k = np.arange(num_points - n, num_points, dtype=np.float64)
    az = (2 * np.pi * k * g) % (2 * np.pi)
    el = np.arcsin(2.0 * k / num_points - 1.0)
    return torch.from_numpy(np.stack([az * 0.5 / np.pi, 1 - 2 * el / np.pi], -1).astype(np.float32))


def sphere_latent(sn):
    """SphereSampler.set_angle (flow.py:62-76) -> [sn,2]."""
    num_points = int(sn // (1 - (1 + 90) / 180))
    g = (np.sqrt(5) - 1.0) / 2.0
    k = np.arange(num_points - sn, num_points, dtype=np.float64)
    phi = (2 * np.pi * k * g) % (2 * np.pi)
    th = np.arcsin(2.0 * k / num_points - 1.0)
    phi = torch.tensor(phi, dtype=torch.float32) / (2 * np.pi)
    th = torch.tensor(th, dtype=torch.float32) / (0.5 * np.pi)
    return torch.stack([phi, th], -1)


_LATENT_ON_DEVICE = {}


def sphere_latent_on(sn, device):
    """sphere_latent(sn) resident on `device`, built once: a host tensor copied per call is a pageable host-to-device copy, i.e. a
    full stream synchronisation at the top of every training step (round 5: it kept a step's forward from being queued under the
    previous step's backward)."""
    key = (int(sn), str(device))
    if key not in _LATENT_ON_DEVICE:
        _LATENT_ON_DEVICE[key] = sphere_latent(sn).to(device)
    return _LATENT_ON_DEVICE[key]


class StageTimer:
    """HIP-event stage timing on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.events = {}
        self.events_overlapped = {}      # work issued on the side stream under main-stream stages: reported apart, not part of the stage sum
        self.units = {}

    class _Ctx:
        def __init__(self, owner, name, overlapped=False):
            self.o, self.name, self.ov = owner, name, overlapped

        def __enter__(self):
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

        def __exit__(self, *a):
            self.e.record()
            (self.o.events_overlapped if self.ov else self.o.events).setdefault(self.name, []).append((self.s, self.e))

    def stage(self, name, overlapped=False):
        """Events are recorded on the CURRENT stream: an overlapped stage is entered inside `with torch.cuda.stream(side)`."""
        return StageTimer._Ctx(self, name, overlapped)

    def add_units(self, name, n):
        self.units[name] = self.units.get(name, 0) + int(n)

    def summary(self):
        """-> {stage: (total_ms, launches)} (call after torch.cuda.synchronize())."""
        return {k: (sum(s.elapsed_time(e) for s, e in v), len(v)) for k, v in self.events.items()}

    def summary_overlapped(self):
        return {k: (sum(s.elapsed_time(e) for s, e in v), len(v)) for k, v in self.events_overlapped.items()}


class _NoTimer:
    class _C:
        def __enter__(self):
            return None

        def __exit__(self, *a):
            return False

    def stage(self, name, overlapped=False):
        return _NoTimer._C()

    def add_units(self, name, n):
        pass


class ShadeOutputs(dict):
    """Output dict of MCShader.shade.  `specular_rays_id` (fields.py:1209-1212: the point index of every unmasked specular
    sample) has a data-dependent length, so building it forces a device->host sync; it is derived from `specular_mask` on
    first access instead of on every call (the eval integral never reads it)."""

    _PER_RAY = ("dirs", "wgt", "hit", "live", "hit_lights", "depth", "inters", "human_hw")
    _LATE = ("colors", "diffuse_lin", "specular_lin")
    _pending = None      # the side stream the reduction of this call was issued on (MCShader.overlap_reduce), until its results are first read

    def __getitem__(self, key):
        if self._pending is not None and key in self._LATE:
            # the per-pixel sums were issued on the side stream so that the NEXT batch's sampling can start under them: whoever reads
            # them first makes its stream wait for that work
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._pending)
            for k in self._LATE:
                dict.__getitem__(self, k).record_stream(cur)
            self._pending = None
        return dict.__getitem__(self, key)

    def __missing__(self, key):
        if key == "hit":           # a ray hit iff its depth is below the traversal's miss value
            v = self["depth"] < ops.MISS_DEPTH
            self[key] = v
            return v
        if key in self._PER_RAY and "_pos_" + key in self:
            # MCShader.shade keeps a point's rays in TRAVERSAL order (row j = slot slot_of_pos[j]); callers that read a per-ray array
            # get it in slot order, gathered on first access (the throughput path never asks)
            v = self["_pos_" + key].index_select(1, self["_pos_of_slot"])
            self[key] = v
            return v
        if key == "lights":     # [pn,T,3] light of every ray (fields.py:951-975): hit rows from the inner-light net, the rest env light
            hit = self["hit"]
            pn, T = hit.shape
            if self["_env"] is None:      # outer_light_version='direction': the outer net wrote the rows of the rays that missed
                self[key] = self["hit_lights"]
                return self[key]
            env = ops.cube_lookup(self["_env"], self["dirs"].reshape(-1, 3), apply_exp=True, depth=self["depth"].reshape(-1), near_eps=1e-5)
            lights = torch.where(hit.reshape(-1, 1), self["hit_lights"].reshape(-1, 3), env).reshape(pn, T, 3)
            self[key] = lights
            return lights
        if key == "inter":      # fields.py:1226,1251: get_lights' intersection rows of the unmasked specular rays [M,3] (hit rows only are
            nd = self["n_diffuse"]                     # meaningful: what the third-party tracer leaves in a missing ray's row is unpinned)
            v = self["inters"][:, nd:][self["specular_mask"].bool()]
            self[key] = v
            return v
        if key == "human_lights":       # fields.py:1226,1241: human_lights * human_weights of the unmasked specular rays that MISSED [n,3]
            nd = self["n_diffuse"]
            sel = self["specular_mask"].bool() & ~self["hit"][:, nd:]
            v = self["human_hw"][:, nd:][sel] if ("human_hw" in self or "_pos_human_hw" in self) else \
                torch.zeros(int(sel.sum()), 3, device=sel.device)
            self[key] = v
            return v
        if key != "specular_rays_id":
            raise KeyError(key)
        smask = self["specular_mask"]
        pn, ss = smask.shape
        rid = torch.arange(pn, device=smask.device)[:, None].expand(pn, ss)[smask.bool()]      # (the kernel writes the mask as bytes)
        self[key] = rid
        return rid


class LazyOutputs(dict):
    """An output dict whose data-dependent-length entries (`inter`, `human_lights`: fields.py:1241,1251 -- nothing outside
    shade_mixed reads them) are built on first access: building them costs a device -> host sync per call."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = {}

    def set_lazy(self, key, fn):
        self._lazy[key] = fn

    def set_lazy_group(self, keys, fn):
        """`fn()` -> a dict holding every key of `keys`, evaluated once on the first access of any of them."""
        keys = tuple(keys)

        def take(key):
            def build():
                d = fn()
                for k in keys:
                    self._lazy.pop(k, None)
                    if k != key and not dict.__contains__(self, k):      # an entry assigned explicitly since stays
                        dict.__setitem__(self, k, d[k])
                return d[key]
            return build
        for k in keys:
            self._lazy[k] = take(k)

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)          # an explicit entry replaces a pending one
        dict.__setitem__(self, key, value)

    def __missing__(self, key):
        if key in self._lazy:
            v = self._lazy.pop(key)()
            self[key] = v
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self):
        return list(dict.keys(self)) + list(self._lazy)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)

    def items(self):
        for k in list(self._lazy):
            self[k]
        return dict.items(self)

    def values(self):
        for k in list(self._lazy):
            self[k]
        return dict.values(self)


def aux_from_stats(aux, nd, ns, metallic, specular_lin, diffuse_lin, diffuse_sample_num):
    """The auxiliary outputs of shade_mixed (fields.py:1228-1256, :1288-1291) from tf_shade_reduce_aux's per-point statistics
    `aux` [pn,16]: light / colour maps, approximate_light, visibility, indirect light and the three variance figures."""
    from .autograd import linear_to_srgb
    c01s = lambda t: linear_to_srgb(t, clamp01=True)
    dmean = aux[:, 0:3] / nd
    spec_color = c01s(specular_lin)
    n, mean, m2 = aux[:, 10:11], aux[:, 11:12], aux[:, 12:13]
    sg, sg2 = n * mean, m2 + n * mean * mean                       # sum g, sum g^2 over the unmasked specular rays
    n64, mean64, m264 = n.double(), mean.double(), m2.double()
    N = n64.sum()
    mu = (n64 * mean64).sum() / N.clamp_min(1.0)
    var_all = ((m264.sum() + (n64 * (mean64 - mu) ** 2).sum()) / (N - 1.0)).float()     # torch.var(unbiased) over all M unmasked rays (:1289)
    return {"diffuse_light": c01s(dmean), "specular_light": c01s(aux[:, 3:6] / ns),
            "diffuse_color": c01s(diffuse_lin), "specular_color": spec_color,
            # (:1248 adds the ALREADY sRGB-encoded, clamped specular colour to the linear diffuse term: reproduced as written)
            "approximate_light": c01s((1 - metallic) * dmean + spec_color),
            "visibility": 1 - aux[:, 9:10] / ns, "indirect_light": aux[:, 6:9] / ns,
            "variance": var_all,
            "variance_diffuse_vis": aux[:, 14:15] / max(nd - 1, 1) / diffuse_sample_num,
            "variance_specular_vis": (sg2 / ns - (sg / ns) ** 2) / ns}


def aux_outputs(out, diffuse_sample_num=512):
    """The auxiliary per-point outputs of shade_mixed (fields.py:1232-1256) from a ShadeOutputs dict built with aux=True (the
    statistics come out of the reduction kernel: no [pn,T,3] light array is materialised)."""
    return aux_from_stats(out["_aux"], out["n_diffuse"], out["specular_mask"].shape[1], out["metallic"], out["specular_lin"],
                          out["diffuse_lin"], diffuse_sample_num)


class FlowParams:
    """One TensoFlow (nis planes/lines + nis_mat + 2 coupling nets) resident on the device."""

    def __init__(self, sd, prefix, device, n_levels=3, field_f16=False):
        g = lambda k: sd[prefix + k].to(device).float().contiguous()
        self.planes = [g(f"nis_plane.{i}") for i in range(3)]
        self.lines = [g(f"nis_line.{i}") for i in range(3)]
        self.packed = ops.VmPacked(self.planes, self.lines, n_levels, texel_f16=field_f16)
        self.mat = [(g("nis_mat.0.weight"), g("nis_mat.0.bias")), (g("nis_mat.2.weight"), g("nis_mat.2.bias"))]
        self.nets = [[(g(f"flows.{b}.nn.{l}.weight"), g(f"flows.{b}.nn.{l}.bias")) for l in (1, 3, 5, 7)] for b in range(2)]
        self.cache = ops.PackCache()     # packed coupling-net fragments survive across calls while the weights are unchanged


class MCShader:
    """Eval-mode MCShadingNetwork, human lights off.  outer_light_version follows the state dict: `outer_light.base` = 'envlight' (the
    cube map, looked up inside the reduction), `outer_light.0.*` = 'direction' (fields.py:716-718, 913-916: a 72-256-256-256-3 net on
    the IDE of the ray direction, evaluated on the rays that missed by tf_outer_light_indexed_fwd; configs/mat/syn/{lego,armadillo,
    horse}.yaml)."""

    def __init__(self, sd, vertices, triangles, aabb, unit_size, device="cuda", n_fixed_diffuse=512,
                 exp_max=5.0, flow_suffix="_copy", precision=ops.PREC_F16X3, n_fixed_specular=256, bvh=None, field_f16=False,
                 light_exp_max=5.0, inner_precision=None, use_half=(True, True), flow_ablate=(False, False), geometry_type="schlick"):
        self.device = device
        if geometry_type not in ("schlick", "ggx_smith"):           # fields.py:1026-1033
            raise NotImplementedError(f"geometry_type {geometry_type!r}: 'schlick' or 'ggx_smith'")
        self.smith = geometry_type == "ggx_smith"
        # cfg disable_tensorial / disable_reflected (fields.py:665-666 -> flow.py:807-812): the flows' condition rows with the tensorial
        # feature (columns 0..15) / the view-angle embedding (16..29) zeroed
        self.flow_ablate = tuple(bool(v) for v in flow_ablate)
        # cfg use_half_diffuse / use_half_specular (fields.py:661-662, :1084, :1163): True (the default) = the flows sample the HALF vector
        self.whole = (not use_half[0], not use_half[1])
        self.precision = precision      # matrix-core arithmetic of the decoders (ops.PREC_F32 = exact fp32 MFMA)
        # Inner-light decoder (123-256-256-256-3).  Library default: ops.PREC_F16X3 -- every operand split hi + lo, the arithmetic of the
        # flow nets and the per-point nets (fp32-grade: 22 significant bits per operand).  ops.PREC_F16X2 is an explicit opt-in
        # (`inner_precision=` here, `--inner-precision f16x2` in bench.py, which labels its line accordingly): weights split hi + lo, the
        # ACTIVATIONS rounded to f16 once per layer (two MFMAs per product term, 128-ray passes of the staggered kernel).  Measured
        # against an fp64 evaluation of the net (tools/exp_il_precision.py, 262 k rays): per ray its worst case and 99.9th percentile
        # sit below those of the reference's own fp32 arithmetic, its rms is 2.5 x the reference's, every golden holds at 1e-4 per pixel
        # (tests/test_gpu_parity.py::test_inner_light_operand_modes_on_trained_like_net) -- tolerance-meeting, but narrower than the
        # reference's fp32, hence never what a caller gets without asking.  ops.PREC_F16 (weights rounded as well) likewise.
        self.inner_precision = ops.PREC_F16X3 if inner_precision is None else int(inner_precision)
        self.cull_dead_rays = True      # skip BVH + inner light for rays whose weight is exactly 0 (result unchanged)
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32)
        self.unit = float(unit_size)
        self.exp_max = exp_max
        g = lambda k: sd[k].to(device).float().contiguous()
        self.mat_planes = [g(f"mat_plane.{i}") for i in range(3)]
        self.mat_lines = [g(f"mat_line.{i}") for i in range(3)]
        # field_f16: BASELINE configs[4] ("fp16 field + flow") -- the material and flow VM pyramids hold halves (ops.VmPacked)
        self.field_f16 = bool(field_f16)
        self.mat_packed = ops.VmPacked(self.mat_planes, self.mat_lines, 3, texel_f16=field_f16)
        sdd = {k: v.to(device).float() for k, v in sd.items() if v.is_floating_point() and ("predictor" in k or "inner_light" in k or
                                                                                          k.startswith("outer_light.") or k.startswith("human_light."))}
        self.pred = {name: [(wn_weight(sdd, f"{name}_predictor.{i}").contiguous(), sdd[f"{name}_predictor.{i}.bias"]) for i in (0, 2)]
                     for name in ("metallic", "roughness", "albedo")}
        self.inner = [(wn_weight(sdd, f"inner_light.{i}").contiguous(), sdd[f"inner_light.{i}.bias"].contiguous()) for i in (0, 2, 4, 6)]
        self.light_exp_max = float(light_exp_max)
        self.outer_sphere, self.human = False, None
        if "outer_light.base" in sd:
            self.env, self.outer = g("outer_light.base"), None
        elif "outer_light.0.bias" in sd and tuple(wn_weight(sdd, "outer_light.0").shape) in ((256, 72), (256, 144)):
            self.env = None
            self.outer = [(wn_weight(sdd, f"outer_light.{i}").contiguous(), sdd[f"outer_light.{i}.bias"].contiguous()) for i in (0, 2, 4, 6)]
            self.outer_cache = ops.PackCache()
            # 144 inputs = 'sphere_direction' (fields.py:917-928: IDE of the direction | IDE of the point where the ray leaves the unit
            # sphere; configs/mat/custom/*): evaluated as a composition (encodings in torch, dense layers on tf_linear_fwd), not fused
            self.outer_sphere = self.outer[0][0].shape[1] == 144
        else:
            raise NotImplementedError("outer light: a cube map (`outer_light.base`, 'envlight') or the 72- / 144-input net of 'direction' / "
                                      "'sphere_direction'")
        if "human_light.0.bias" in sd:
            # human_lights (fields.py:727-729, 935-949; configs/mat/custom/*): light reflected off the photo capturer, a 24-256-256-256-4
            # net on the positional encoding of where a missing ray meets the capturer's plane; blended into the outer light
            self.human = [(wn_weight(sdd, f"human_light.{i}").contiguous(), sdd[f"human_light.{i}.bias"].contiguous()) for i in (0, 2, 4, 6)]
            if self.outer is None:
                raise NotImplementedError("human_lights with the cube-map outer light: no shipped config combines them")
        self.flow_d = FlowParams(sd, f"flow_diffuse{flow_suffix}.", device, field_f16=field_f16)
        self.flow_s = FlowParams(sd, f"flow_specular{flow_suffix}.", device, field_f16=field_f16)
        self.inner_cache = ops.PackCache()
        self.bvh = bvh if bvh is not None else ops.Bvh(vertices, triangles, device)   # `bvh`: reuse an uploaded tree (same mesh)
        self.point_prep = ops.PointPrep(self.mat_packed, self.flow_d.packed, self.flow_s.packed, self.pred,
                                        [self.flow_d.mat, self.flow_s.mat], self.aabb)
        self.fixed_d = fibonacci_samples(n_fixed_diffuse).to(device)
        self.fixed_s = fibonacci_samples(n_fixed_specular).to(device)      # non-NIS pass only (fields.py:739-742)
        self._latent = {}
        self._order = {}
        self.sort_rays = True           # trace each point's rays in direction-sorted order (results unchanged)
        self.sort_origins = False       # hand the points to the traversal in Morton order (results unchanged; measured: no gain, +1 ms of sorting)
        self.overlap_dirs = True        # build the diffuse / fixed direction rows on a second stream under the specular flow's sampling
        self.overlap_reduce = False     # True: issue the per-pixel reduction on the second stream, under the NEXT batch's per-point stage and
                                        # flow sampling.  Measured slower (41.7 vs 40.7 ms per step): its texel gathers slow the flow kernel by
                                        # more (+2.1 ms) than the reduction's own 2.7 ms
        self._side_streams = {}         # per calling stream (two shade() calls may be in flight on two streams: shade_many)
        self._call_streams = []
        self.timer = _NoTimer()
        self._hit_totals = {}           # per calling stream: device-side tallies of the hit rays (no sync)
        self.human_hw = None

    def _side_stream_of(self, cur, dev):
        key = int(cur.cuda_stream)
        if key not in self._side_streams:
            self._side_streams[key] = torch.cuda.Stream(device=dev)
        return self._side_streams[key]

    @property
    def hit_total(self):
        """Sum of the per-stream hit tallies (a device tensor; synchronise before reading it from the host), None before any call."""
        if not self._hit_totals:
            return None
        vals = list(self._hit_totals.values())
        cur = torch.cuda.current_stream()
        for s in self._call_streams:
            cur.wait_stream(s)
        tot = vals[0].clone()
        for v in vals[1:]:
            tot = tot + v
        return tot

    @hit_total.setter
    def hit_total(self, v):
        if v is not None:
            raise ValueError("hit_total can only be reset (None)")
        self._hit_totals = {}

    @torch.no_grad()
    def shade_many(self, pts, view_dirs, normals, sn_diffuse, sn_specular, chunk, n_streams=1, keep=("colors",)):
        """The chunk loop of a frame / a batch (materialRenderer.py:705-709 shades 512 rays per pass, strictly one after another), with
        `n_streams` shade() calls in flight on as many HIP streams (every stream has its own workspaces: ops.PackCache, side streams,
        hit tallies).  Results are bit-identical to the serial loop (tests/test_gpu_determinism.py) -- after round 5 found and removed
        the one thing that was not: hipcc's packed-fp32 form of `view_angles_kernel` (csrc/view_angles.hip) misbehaved on a few lanes
        whenever another kernel's waves were resident beside it.  Measured gain: none (183.1 against 182.2 ms per 2^20 points: the stage
        kernels cannot hide under one another), so n_streams = 1 stays the default.
        -> list of dicts holding the `keep` entries per chunk."""
        cur = torch.cuda.current_stream()
        while len(self._call_streams) < n_streams:
            self._call_streams.append(torch.cuda.Stream(device=pts.device))
        outs = []
        for i, c0 in enumerate(range(0, pts.shape[0], chunk)):
            s = self._call_streams[i % n_streams] if n_streams > 1 else cur
            if n_streams > 1:
                s.wait_stream(cur)
            with torch.cuda.stream(s):
                o = self.shade(pts[c0:c0 + chunk], view_dirs[c0:c0 + chunk], normals[c0:c0 + chunk], sn_diffuse, sn_specular)
                outs.append({k: o[k] for k in keep if k != "_live_count"})
                if "_live_count" in keep:          # the number of rays the traversal was handed (weight != 0): a device scalar
                    outs[-1]["_live_count"] = o["_pos_live" if "_pos_live" in o else "live"].sum(dtype=torch.int64)
        if n_streams > 1:
            for s in self._call_streams[:n_streams]:
                cur.wait_stream(s)
            for o in outs:
                for v in o.values():
                    v.record_stream(cur)
        return outs

    def latent(self, sn):
        if sn not in self._latent:
            self._latent[sn] = sphere_latent(sn).to(self.device)
        return self._latent[sn]

    def slot_order(self, sn_d, sn_s):
        """Traversal order of a point's T = sn_d + n_fixed + sn_s secondary rays (tf_bvh_trace slot_order): inside each of
        the three direction sets the slots are sorted along a Morton curve over their (azimuth, elevation) cell, so the 64
        rays of a wavefront point the same way.  The sets themselves are Fibonacci spirals -- neighbouring slots are ~222
        degrees apart in azimuth.  (The flow-warped sets are sorted by their latent position: the warp is smooth.)"""
        key = (sn_d, sn_s)
        if key not in self._order:
            def morton(a):                      # a [n,2] in [0,1]^2 -> sort index
                q = np.clip((a * 16).astype(np.int64), 0, 15)
                code = np.zeros(len(a), np.int64)
                for b in range(4):
                    code |= ((q[:, 0] >> b) & 1) << (2 * b + 1) | ((q[:, 1] >> b) & 1) << (2 * b)
                return np.argsort(code, kind="stable")
            nf = self.fixed_d.shape[0]
            parts = [morton(sphere_latent(sn_d).numpy()), sn_d + morton(self.fixed_d.cpu().numpy()),
                     sn_d + nf + morton(sphere_latent(sn_s).numpy())]
            self._order[key] = torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(self.device)
        return self._order[key]

    def pos_of_slot(self, sn_d, sn_s):
        """Inverse of slot_order (int64, for index_select): the row of a point's per-ray arrays that holds slot s."""
        key = ("inv", sn_d, sn_s)
        if key not in self._order:
            self._order[key] = torch.argsort(self.slot_order(sn_d, sn_s).long())
        return self._order[key]

    @staticmethod
    def _net4(weights, x, exp_max):
        """A 4-layer predictor (ReLU x 3, exp(min(., exp_max))) on the dense-layer kernels."""
        for l, (W, b) in enumerate(weights):
            x = ops.linear_fwd(x, W, b, ops.ACT_RELU if l < 3 else ops.ACT_EXP_CLAMP, exp_max)
        return x

    def miss_lights_composed(self, origins, dirs, poses):
        """predict_outer_lights('sphere_direction') / get_human_light + their blend (fields.py:913-949, 962-968) for rays that missed:
        origins, dirs [n,3], poses [n,3,4] or None -> lights [n,3]."""
        from .encodings import ide5
        zero = torch.zeros(dirs.shape[0], 1, device=dirs.device)
        enc = ide5(dirs, zero, wide=True)
        if self.outer_sphere:
            o = origins.clone()
            far = o.norm(dim=-1) > 0.999
            o[far] = o[far] * 0.999                                        # (shrink this point a little bit, :922-924)
            dtx = (o * dirs).sum(-1, keepdim=True)
            dist = -dtx + torch.sqrt(dtx ** 2 - (o ** 2).sum(-1, keepdim=True) + 1 + 1e-6)       # get_sphere_intersection
            enc = torch.cat([enc, ide5(o + dirs * dist, zero, wide=True)], -1)
        outer = self._net4(self.outer, enc.contiguous(), self.light_exp_max)
        self._last_hlhw = None
        if self.human is None or poses is None:        # (no poses: the outer net alone -- MCShader.lights() on bare rays)
            return outer
        R, t = poses[:, :, :3], poses[:, :, 3]
        p_ = torch.einsum("nij,nj->ni", R, origins) + t
        d_ = torch.einsum("nij,nj->ni", R, dirs)
        hits = d_[:, 2].abs() > 1e-4
        dz = torch.where(hits, d_[:, 2], torch.full_like(d_[:, 2], 1e-4))
        d_ = torch.cat([d_[:, :2], dz[:, None]], -1)                       # (the reference overwrites the z of the view it divides by)
        dist = -p_[:, 2] / dz
        mean = (p_ + dist[:, None] * d_)[:, :2] * 0.3
        hits = (hits & (mean.norm(dim=-1) < 1.5) & (dist > 0)).float()[:, None]
        mean = mean * hits
        scaled = (mean[:, None, :] * (2.0 ** torch.arange(6, device=mean.device))[:, None]).reshape(-1, 12)       # IPE(mean, 0, 0, 6)
        pe = torch.sin(torch.cat([scaled, scaled + 0.5 * math.pi], -1))
        h = self._net4(self.human, pe.contiguous(), 0.0) * hits            # ExpActivation(max_light = 0): at most 1
        hl, hw = h[:, :3], h[:, 3:].clamp(0.0, 1.0)
        self._last_hlhw = hl * hw
        return outer * (1 - hw) + hl * hw

    def trace_and_inner(self, pts_rep, dirs, live=None, slot_order=None, origin_order=None, poses=None):
        """Hit branch of get_lights (fields.py:951-975): BVH visibility + inner-light MLP on the rays that hit.
        -> hit_lights [M,3] (rows of rays that hit; the others are uninitialised), hit = None (a ray hit iff depth < ops.MISS_DEPTH:
        the traversal stores no separate flag byte -- 0.6 ms of scattered one-byte stores per 201 M rays), depth [M], inters [M,3]."""
        T = self.timer
        with T.stage("bvh_trace"):
            # the hit point / normal rows are only read through the compacted hit list below
            inters, nrm, depth, hit = self.bvh.trace(pts_rep, dirs, 1e-5, 2 * self.unit, live=live, slot_order=slot_order,
                                                     hit_rows_only=True, origin_order=origin_order, want_hit=False)
        with T.stage("hit_compaction"):
            idx, count = ops.compact_below(depth, ops.MISS_DEPTH)
        with T.stage("inner_light"):
            # 'direction': rows of culled rays are read by nobody but `lights` consumers (aux maps): zero, not uninitialised
            hit_lights = torch.empty_like(dirs) if self.outer is None else torch.zeros_like(dirs)
            ops.inner_light_indexed(self.inner, inters, dirs, nrm, idx, count, depth, hit_lights, near_eps=1e-5,
                                    exp_max=self.exp_max, precision=self.inner_precision if self.precision != ops.PREC_F32 else ops.PREC_F32,
                                    cache=self.inner_cache)
        if self.outer is not None:
            with T.stage("outer_light"):
                # miss branch of get_lights (fields.py:962-968) with predict_outer_lights('direction'): the same staggered kernel on the
                # (live) rays that missed; their near mask is 1 (depth = the traversal's miss value)
                miss = (depth >= ops.MISS_DEPTH).to(torch.uint8)
                if live is not None:
                    miss &= live.reshape(-1).to(torch.uint8)
                idx_m, count_m = ops.compact_mask(miss)
                if not self.outer_sphere and self.human is None:
                    ops.outer_light_indexed(self.outer, dirs, idx_m, count_m, hit_lights, exp_max=self.light_exp_max, cache=self.outer_cache,
                                            precision=self.inner_precision if self.inner_precision in (ops.PREC_F16X3, ops.PREC_F16X2) else ops.PREC_F16X3)
                else:
                    # composed variants (configs/mat/custom): one host sync for the count, slices of 2^21 rays to bound the encodings
                    n, T = int(count_m), dirs.shape[0] // pts_rep.shape[0]
                    # ray r belongs to origin r // T: `dirs` is [pn, T] row-major whatever `origin_order` says (that argument only
                    # changes the order in which the traversal's waves CLAIM the origins; no row moves)
                    assert dirs.shape[0] == pts_rep.shape[0] * T, (dirs.shape, pts_rep.shape)
                    self.human_hw = torch.zeros_like(dirs) if (self.human is not None and poses is not None) else None
                    for c0 in range(0, n, 1 << 21):
                        ids = idx_m[c0:min(c0 + (1 << 21), n)]
                        org = ids // T if T > 1 else ids
                        hit_lights[ids] = self.miss_lights_composed(pts_rep[org], dirs[ids], poses[org] if poses is not None else None)
                        if self.human_hw is not None and self._last_hlhw is not None:
                            self.human_hw[ids] = self._last_hlhw           # human_lights * human_weights (get_lights' 2nd value, :975)
        key = int(torch.cuda.current_stream().cuda_stream)          # device-side tally (no sync), one per calling stream
        self._hit_totals[key] = count if key not in self._hit_totals else self._hit_totals[key] + count
        return hit_lights, hit, depth, inters

    def lights(self, pts_rep, dirs, live=None, slot_order=None):
        """get_lights (fields.py:951-975) for every ray: pts_rep [M,3] (or [M // T, 3]: T consecutive rays per origin),
        dirs [M,3] -> lights [M,3], hit [M] bool, inters.  (shade() does not materialise this array: the miss branch is
        evaluated inside the reduction, tf_shade_reduce_env.)"""
        hit_lights, _, depth, inters = self.trace_and_inner(pts_rep, dirs, live=live, slot_order=slot_order)
        hit = depth < ops.MISS_DEPTH
        if self.outer is not None:
            return hit_lights, hit, inters
        lights = ops.cube_lookup(self.env, dirs, apply_exp=True, depth=depth, near_eps=1e-5)
        lights = torch.where(hit[:, None], hit_lights, lights)
        return lights, hit, inters

    @torch.no_grad()
    def shade_fixed(self, pts, view_dirs, normals, human_poses=None, aux=False):
        """The non-NIS pass of shade_mixed (nis_sample=False, fields.py:1075-1235 with the `else` samplers): the fixed cosine set
        for the diffuse lobe and the fixed GGX-warped set for the specular lobe (sample_diffuse_directions /
        sample_specular_directions, :824-903).  Same output dict as `shade` (without the flow arrays)."""
        pts = pts.to(self.device).float().contiguous()
        pn = pts.shape[0]
        va = ops.view_angles(normals, view_dirs)
        metallic, rough, albedo, _, _ = self.point_prep(pts, va)
        dirs, wgt, smask, live = ops.shade_dirs_fixed(normals, view_dirs, metallic, rough, albedo, self.fixed_d, self.fixed_s, smith=self.smith)
        T, nd, ns = dirs.shape[1], self.fixed_d.shape[0], self.fixed_s.shape[0]
        # aux: the unweighted maps average over EVERY ray (zero weight or not), so nothing is culled for such a call
        hit_lights, hit, depth, inters = self.trace_and_inner(pts, dirs.reshape(-1, 3), live=live if (self.cull_dead_rays and not aux) else None,
                                                              poses=human_poses)
        if aux:
            colors, dl, sl, stats = ops.shade_reduce_aux(wgt, smask, nd, ns, dirs=dirs, depth=depth, hit_lights=hit_lights, env_base=self.env)
        else:
            colors, dl, sl = ops.shade_reduce_env(wgt, dirs, depth, None, hit_lights, self.env, nd, ns)
        out = ShadeOutputs(colors=colors, diffuse_lin=dl, specular_lin=sl, metallic=metallic, roughness=rough, albedo=albedo,
                           specular_mask=smask, live=live, view_angles=va, dirs=dirs, wgt=wgt, inters=inters.reshape(pn, T, 3),
                           hit_lights=hit_lights.reshape(pn, T, 3), depth=depth.reshape(pn, T), _env=self.env, n_diffuse=nd)
        if aux:
            out["_aux"] = stats
        if self.human_hw is not None and human_poses is not None:
            out["human_hw"] = self.human_hw.reshape(pn, T, 3)
        return out

    @torch.no_grad()
    def shade(self, pts, view_dirs, normals, sn_diffuse, sn_specular, jitter_d=None, jitter_s=None, human_poses=None, aux=False):
        """-> dict(colors [pn,3], specular_mask, specular_rays_id, diffuse_lin, specular_lin, materials...)
        One call at a time per shader: the kernels' workspaces (packed weights, the flows' per-point rows, the traversal's work
        counters) are per device, not per call -- two shade() calls in flight on different streams would share them (measured as
        well: two batches in flight are 2x SLOWER, the persistent traversal kernels of both cannot be resident together)."""
        pts = pts.to(self.device).float().contiguous()
        pn = pts.shape[0]
        tm = self.timer
        with tm.stage("point_prep"):
            # materials + both flow condition rows: one fused launch (tf_point_fwd) after the view-angle kernel
            va = ops.view_angles(normals, view_dirs)
            metallic, rough, albedo, cond_d, cond_s = self.point_prep(pts, va)
            if any(self.flow_ablate):
                for cnd in (cond_d, cond_s):
                    if self.flow_ablate[0]:
                        cnd[:, :16] = 0
                    if self.flow_ablate[1]:
                        cnd[:, 16:30] = 0
        # a point's rays are STORED in traversal order (row j holds slot order[j]): the traversal then reads and writes consecutive
        # rows from consecutive lanes instead of going through the permutation for every ray; only the kernels that need to know
        # WHICH sample a row is (direction construction, the lobe split of the reduction) take the permutation
        order = self.slot_order(sn_diffuse, sn_specular) if self.sort_rays else None
        nf = self.fixed_d.shape[0]
        with tm.stage("flow_sample"):
            ang_d, lq_d = ops.flow_sample(self.flow_d.nets, cond_d, self.latent(sn_diffuse), jitter_d, precision=self.precision,
                                          cache=self.flow_d.cache)
        if self.overlap_dirs:
            # the rows of the diffuse flow set and of the fixed set (5/6 of a point's rows: an HBM write stream) are built on a second
            # HIP stream WHILE the specular flow is being sampled (matrix-core / vector-issue bound, HBM idle); the specular rows follow
            T = sn_diffuse + nf + sn_specular
            dev = pts.device
            bufs = (torch.empty(pn, T, 3, dtype=torch.float32, device=dev), torch.empty(pn, T, 3, dtype=torch.float32, device=dev),
                    torch.empty(pn, sn_specular, dtype=torch.bool, device=dev), torch.empty(pn, T, dtype=torch.uint8, device=dev))
            cur = torch.cuda.current_stream()
            side = self._side_stream_of(cur, dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                with tm.stage("shade_dirs (diffuse + fixed rows)", overlapped=True):
                    ops.shade_dirs(normals, view_dirs, metallic, rough, albedo, ang_d, lq_d, self.fixed_d, (sn_specular,), None,
                                   slot_of_pos=order, rows=(0, sn_diffuse + nf), out=bufs, whole=self.whole, smith=self.smith)
            with tm.stage("flow_sample"):
                ang_s, lq_s = ops.flow_sample(self.flow_s.nets, cond_s, self.latent(sn_specular), jitter_s, precision=self.precision,
                                              cache=self.flow_s.cache)
            with tm.stage("shade_dirs"):
                cur.wait_stream(side)
                dirs, wgt, smask, live = ops.shade_dirs(normals, view_dirs, metallic, rough, albedo, ang_d, lq_d, self.fixed_d, ang_s, lq_s,
                                                        slot_of_pos=order, rows=(sn_diffuse + nf, sn_specular), out=bufs, whole=self.whole, smith=self.smith)
        else:
            with tm.stage("flow_sample"):
                ang_s, lq_s = ops.flow_sample(self.flow_s.nets, cond_s, self.latent(sn_specular), jitter_s, precision=self.precision,
                                              cache=self.flow_s.cache)
            with tm.stage("shade_dirs"):
                dirs, wgt, smask, live = ops.shade_dirs(normals, view_dirs, metallic, rough, albedo, ang_d, lq_d, self.fixed_d, ang_s, lq_s,
                                                        slot_of_pos=order, whole=self.whole, smith=self.smith)
        tm.add_units("flow_sample", pn * (sn_diffuse + sn_specular))
        T = dirs.shape[1]
        # the T secondary rays of a point share its origin row (tf_bvh_trace rays_per_origin = T): pts[:,None].expand is never built
        # origins are handed to the traversal in Morton order (each XCD then works on one contiguous eighth of the scene); nothing
        # is moved: only the order in which the persistent waves claim the points changes
        with tm.stage("point_prep"):
            oorder = ops.morton_order(pts, self.aabb) if self.sort_origins and T >= 64 else None
        # aux=True (the rest of shade_mixed's output dict, fields.py:1232-1256): every ray's light enters the unweighted maps, so the
        # zero-weight culling is off for such a call and the reduction is tf_shade_reduce_aux
        hit_lights, hit, depth, inters = self.trace_and_inner(pts, dirs.reshape(-1, 3), live=live if (self.cull_dead_rays and not aux) else None,
                                                              origin_order=oorder, poses=human_poses)
        n_diff = sn_diffuse + self.fixed_d.shape[0]
        pending = None
        stats = None
        if aux:
            with tm.stage("shade_reduce"):
                colors, dl, sl, stats = ops.shade_reduce_aux(wgt, smask, n_diff, sn_specular, dirs=dirs, depth=depth, hit_lights=hit_lights,
                                                             env_base=self.env, slot_of_pos=order)
        elif self.overlap_reduce:
            # the reduction (texel gathers of the environment light: latency bound, matrix cores idle) goes to the side stream: it runs
            # under the per-point stage and the flow sampling of the NEXT batch unless the caller reads the colours first
            # (ShadeOutputs.__getitem__ waits).  Its inputs are kept from the allocator until that work is done (record_stream).
            pending = self._side_stream_of(torch.cuda.current_stream(), pts.device)
            pending.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(pending):
                with tm.stage("shade_reduce", overlapped=True):
                    colors, dl, sl = ops.shade_reduce_env(wgt, dirs, depth, None, hit_lights, self.env, n_diff, sn_specular, slot_of_pos=order)
            for t in (wgt, dirs, depth, hit_lights):
                t.record_stream(pending)
        else:
            with tm.stage("shade_reduce"):
                # environment light of the rays that missed is evaluated inside the reduction (no [pn,T,3] light array)
                colors, dl, sl = ops.shade_reduce_env(wgt, dirs, depth, None, hit_lights, self.env, n_diff, sn_specular, slot_of_pos=order)
        out = ShadeOutputs(colors=colors, diffuse_lin=dl, specular_lin=sl, metallic=metallic, roughness=rough, albedo=albedo,
                           specular_mask=smask, view_angles=va, diffuse_angles=ang_d, diffuse_logq=lq_d, specular_angles=ang_s,
                           specular_logq=lq_s, _env=self.env, n_diffuse=n_diff)
        per_ray = dict(live=live, dirs=dirs, wgt=wgt, hit_lights=hit_lights.reshape(pn, T, 3), depth=depth.reshape(pn, T),
                       inters=inters.reshape(pn, T, 3))
        if self.human_hw is not None and human_poses is not None:
            per_ray["human_hw"] = self.human_hw.reshape(pn, T, 3)
        if stats is not None:
            out["_aux"] = stats
        if order is None:
            out.update(per_ray)
        else:
            out.update({"_pos_" + k: v for k, v in per_ray.items()})
            out["_pos_of_slot"] = self.pos_of_slot(sn_diffuse, sn_specular)
        out._pending = pending
        return out
