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
