"""The drop-in modules keep the reference's state_dict keys and shapes (checked against the goldens, which hold the
reference modules' own state_dicts) -- CPU-only construction, no kernel is launched."""
import torch

from conftest import AABB


def test_tensoflow_state_dict_keys(golden):
    from tensoflow_amd.network.flow import TensoFlow
    g = golden("tensoflow_r32")
    m = TensoFlow(2, AABB, device="cpu", gridSize=[32, 32, 32])
    sd = m.state_dict()
    assert set(sd) == set(g.sd), set(sd) ^ set(g.sd)
    for k, v in g.sd.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict(g.sd)                                     # a reference checkpoint loads unchanged
    groups = m.get_optparam_groups(0.01, 0.001)
    assert [g_["lr"] for g_ in groups] == [0.01, 0.01, 0.001, 0.001]


def test_tensosdf_state_dict_keys(golden):
    from tensoflow_amd.network.fields import TensoSDF
    g = golden("tensosdf_r32_l3")
    m = TensoSDF(torch.tensor([32, 32, 32]), AABB, device="cpu", init_n_levels=3)
    assert set(m.state_dict()) == set(g.sd)                       # incl. the Gaussian-blur buffers of grid_gaussian_loss
    for k in g.sd:
        assert tuple(m.state_dict()[k].shape) == tuple(g.sd[k].shape), k
    m.load_state_dict(g.sd)
    assert torch.allclose(m.units, g["units"])


def test_fused_forward_refuses_autograd():
    import pytest
    from tensoflow_amd.network.flow import TensoFlow
    m = TensoFlow(2, AABB, device="cpu", gridSize=[8, 8, 8])
    with pytest.raises(RuntimeError, match="no backward yet"):
        m.sample(torch.zeros(2, 3), torch.zeros(2, 2), torch.zeros(2, 1), 8, return_jacobian=True)   # sampling is a frozen-copy op
