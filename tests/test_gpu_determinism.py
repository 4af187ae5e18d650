"""Run-to-run determinism of the fused kernels: a hardware hazard the compiler misses (e.g. a packed-FP32 read of a matrix-core
result, found and removed in flow.hip) shows up as differing bits between two identical launches long before it trips a
tolerance test.  The forward kernels use no floating-point atomics, so identical inputs must give identical bits."""
import pytest
import torch

from conftest import AABB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected but no GPU is visible")
    return torch.device("cuda:0")


def _same(fn, n=3):
    ref = [t.clone() for t in fn()]
    for _ in range(n - 1):
        for a, b in zip(ref, fn()):
            assert torch.equal(a, b)


def test_flow_and_inner_light_bitwise_repeatable(dev):
    from tensoflow_amd import ops
    from tensoflow_amd.shading import FlowParams, sphere_latent, wn_weight
    from tensoflow_amd.synth import random_mc_state
    sd = random_mc_state(seed=4, R=32, flow_R=32, env_res=8)
    fp = FlowParams(sd, "flow_diffuse_copy.", dev)
    cond = torch.randn(20000, 37, generator=torch.Generator().manual_seed(1)).to(dev)
    lat = sphere_latent(128).to(dev)
    for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16):
        _same(lambda: ops.flow_sample(fp.nets, cond, lat, None, precision=prec))
    W = [(wn_weight(sd, f"inner_light.{i}").to(dev), sd[f"inner_light.{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
    g = torch.Generator().manual_seed(2)
    p, v, n = [torch.randn(100_003, 3, generator=g).to(dev) for _ in range(3)]
    for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16X2, ops.PREC_F16):
        _same(lambda: [ops.inner_light(W, p, v, n, precision=prec)])


def test_march_kernels_bitwise_repeatable(golden, dev):
    from tensoflow_amd import ops
    from tensoflow_amd.march import SdfField
    from tensoflow_amd.shape_shading import ShapeShader
    g = golden("march_r32")
    f = SdfField(g.sd, AABB, [32, 32, 32], 3, device=dev)
    c = lambda k: g[k].to(dev)
    ridx = c("ray_indices")
    for prec in (ops.PREC_F32, ops.PREC_F16X3):
        _same(lambda: [t for t in f.sdf_alpha(c("sample_pts"), c("sample_levels")[:, 0].contiguous(), c("t_ends") - c("t_starts"),
                                              c("dirs")[ridx].contiguous(), 20.0, 0.5, precision=prec) if t is not None])
    sh = ShapeShader(g.sd, [g["env_spec0"], g["env_spec1"], g["env_spec2"]], g["env_diffuse"], g["fg_lut"], device=dev)
    nrm = torch.nn.functional.normalize(c("sa_grad"), dim=-1)
    _same(lambda: list(sh(c("sample_pts"), nrm, (-c("dirs")[ridx]).contiguous(), c("sa_feat"))))


def test_launch_budget_does_not_change_a_bit(dev):
    """tf_set_launch_budget (round 5) only changes how much of a CU the three stage kernels of the integral take -- persistent traversal
    workgroups per CU, waves of the flow kernel's workgroup, one or two teams in the inner-light workgroup -- never a result: depths,
    hit points and normals, flow samples and log-densities, and the inner-light radiance are bit-identical under every budget."""
    import numpy as np
    from tensoflow_amd import ops
    from tensoflow_amd.shading import FlowParams, sphere_latent, wn_weight
    from tensoflow_amd.synth import random_mc_state, sphere_torus_mesh
    sd = random_mc_state(seed=4, R=32, flow_R=32, env_res=8)
    try:
        # traversal: 4 096 origins x 128 rays through the spine kernel
        v, f = sphere_torus_mesh(48, 96, 64, 32)
        bvh = ops.Bvh(v, f, dev)
        g = torch.Generator().manual_seed(7)
        o = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1).mul(0.5).to(dev)
        d = torch.nn.functional.normalize(torch.randn(4096 * 128, 3, generator=g), dim=-1).to(dev)
        ref = None
        for k in (0, 1, 3, 6, 8):
            ops.set_launch_budget(bvh_blocks_per_cu=k)
            pos, nrm, depth, hit = bvh.trace(o, d, 1e-5, 4e-3)
            got = (depth.clone(), hit.clone(), torch.where(hit[:, None], pos, torch.zeros_like(pos)), torch.where(hit[:, None], nrm, torch.zeros_like(nrm)))
            ref = ref or got
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), k
        assert 0.05 < float(ref[1].float().mean()) < 0.95
        # flow sampling and density
        fp = FlowParams(sd, "flow_diffuse_copy.", dev)
        cond = torch.randn(20000, 37, generator=torch.Generator().manual_seed(1)).to(dev)
        lat = sphere_latent(128).to(dev)
        ref = None
        for w in (0, 4, 8, 12):
            ops.set_launch_budget(flow_waves_per_block=w)
            ang, lq = ops.flow_sample(fp.nets, cond, lat, None, precision=ops.PREC_F16X3)
            z, logq = ops.flow_logq(fp.nets, cond, ang, precision=ops.PREC_F16X3)
            got = (ang.clone(), lq.clone(), z.clone(), logq.clone())
            ref = ref or got
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), w
        # inner light through a ragged hit list, both forms of the staggered kernel
        W = [(wn_weight(sd, f"inner_light.{i}").to(dev), sd[f"inner_light.{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
        g = torch.Generator().manual_seed(2)
        n = 100_003
        p, vv, nn = [torch.randn(n, 3, generator=g).to(dev) for _ in range(3)]
        dep = (torch.rand(n, generator=g) + 0.1).to(dev)
        idx = torch.nonzero(torch.rand(n, generator=g) < 0.3)[:, 0].to(dev)
        cnt = torch.tensor([idx.numel() - 5], dtype=torch.int64, device=dev)
        for prec in (ops.PREC_F16X3, ops.PREC_F16X2):
            ref = None
            for t in (0, 1, 2):
                ops.set_launch_budget(inner_teams=t)
                out = torch.zeros(n, 3, device=dev)
                ops.inner_light_indexed(W, p, vv, nn, idx, cnt, dep, out, precision=prec)
                ref = out.clone() if ref is None else ref
                assert torch.equal(ref, out), (prec, t)
            assert float(ref.abs().sum()) > 0
        with pytest.raises(RuntimeError, match="inner_teams"):
            ops.set_launch_budget(inner_teams=3)
    finally:
        ops.set_launch_budget()


def test_two_calls_in_flight_equal_the_serial_loop(dev):
    """MCShader.shade_many with two shade() calls in flight on two HIP streams (own workspaces per stream) returns the colours of the
    serial chunk loop bit for bit.  Round 5 found the one kernel for which that did not hold -- hipcc's packed-fp32 form of
    view_angles_kernel took the other tangent-frame candidate on a few lanes whenever another kernel's waves were resident beside
    it (csrc/view_angles.hip is built without SLP vectorisation since) -- so this test runs calls that START TOGETHER, several times."""
    import bench
    from tensoflow_amd.synth import sphere_surface_points
    sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (64, 128, 96, 48))
    pn, chunk = 98304, 24576
    pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=12)]
    ref = torch.cat([o["colors"] for o in sh.shade_many(pts, view, nrm, 128, 128, chunk, n_streams=1)])
    for _ in range(4):
        got = torch.cat([o["colors"] for o in sh.shade_many(pts, view, nrm, 128, 128, chunk, n_streams=2)])
        assert torch.equal(got, ref)
    assert torch.isfinite(ref).all() and float(ref.std()) > 0.01
