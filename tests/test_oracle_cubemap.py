"""CPU checks of the env-light prefilter oracle (oracle/cubemap.py).  No golden vectors exist for this row (the CUDA plugin
cannot run here): the pins are structural -- adjointness, energy of constant maps, bounds-free equivalence."""
import numpy as np

from oracle import cubemap as oc


def test_texel_dirs_unit_and_faces():
    d = oc.texel_dirs(8)
    assert d.shape == (6, 8, 8, 3)
    assert np.allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)
    # GL face order +x,-x,+y,-y,+z,-z (light_utils.py:24-31)
    for s, (ax, sgn) in enumerate([(0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1)]):
        assert (np.sign(d[s, ..., ax]) == sgn).all()
        assert (np.abs(d[s, ..., ax]) >= np.abs(d[s]).max(-1) - 1e-6).all()


def test_texel_area_asymmetry_and_sum():
    a = oc.texel_area(16)
    # |x-H| indexing of cubemap.cu:17-31: texel H-1 uses the interval [1,2], texel H uses [0,1]
    assert a[8, 8] > a[7, 7]
    assert np.isclose(a[8, 8], np.arctan(1 / 8) ** 2, rtol=1e-5)
    assert oc.texel_area(1)[0, 0] == 1.0


def test_constant_map_specular_is_identity_and_diffuse_scales():
    cub = np.full((6, 16, 16, 3), -0.7, np.float32)
    out = oc.specular(cub, 0.5)
    assert np.allclose(out, -0.7, atol=1e-5)
    dif = oc.diffuse(cub)
    w = oc._diffuse_weights(16).sum(-1)
    assert np.allclose(dif[..., 0].reshape(-1), -0.7 * w, rtol=1e-5)
    assert 0.8 < w.mean() < 1.2          # ~ integral of cos over the hemisphere / pi


def test_adjoints():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((6, 8, 8, 3)).astype(np.float32)
    g = rng.standard_normal((6, 8, 8, 3)).astype(np.float32)
    assert np.isclose((oc.diffuse(x) * g).sum(), (x * oc.diffuse_bwd(g)).sum(), rtol=1e-4)
    for r in (0.3, 1.0):
        assert np.isclose((oc.specular(x, r) * g).sum(), (x * oc.specular_bwd(g, 8, r)).sum(), rtol=1e-3)


def test_ndf_cutoff_monotone():
    cs = [oc.ndf_cutoff(r) for r in (0.08, 0.29, 0.5, 1.0)]
    assert cs[0] > cs[1] > cs[2] > cs[3] > 0.0
    assert abs(cs[3] - np.cos(0.99 * np.pi / 2)) < 1e-4        # uniform NDF at roughness 1


def test_build_mips_shapes_and_mip():
    rng = np.random.default_rng(1)
    base = (np.log(0.5) + 0.5 * rng.standard_normal((6, 32, 32, 3))).astype(np.float32)
    spec, diff = oc.build_mips(base, min_res=8)
    assert [s.shape[1] for s in spec] == [32, 16, 8] and diff.shape == (6, 8, 8, 3)
    m = oc.mip(base)
    assert np.allclose(m[2, 3, 5], base[2, 6:8, 10:12].mean((0, 1)), atol=1e-6)
    g = oc.mip_bwd(m)
    assert g.shape == base.shape
