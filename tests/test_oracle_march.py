"""Oracle ray-march restatement vs golden vectors produced by the imported reference
(ShapeRenderer.sample_ray / compute_sdf_alpha / render_core / ShapeShadingNetwork.forward)."""
import torch

from conftest import AABB, rel_err
from oracle import march as om

GS = torch.tensor([32, 32, 32])


def _env(g):
    return dict(diffuse=g["env_diffuse"], specular=[g["env_spec0"], g["env_spec1"], g["env_spec2"]])


def test_sample_ray_indices_bit_exact(golden):
    g = golden("march_r32")
    t0, t1, ridx = om.sample_ray(g.sd, g["rays_o"], g["dirs"], g["near"], g["far"], g["radiis"], g["rays_cos"],
                                 AABB, GS, 3, float(g["base_radii"]))
    assert torch.equal(ridx, g["ray_indices"])                 # int64, bit-exact
    assert rel_err(t0, g["t_starts"]) < 1e-6 and rel_err(t1, g["t_ends"]) < 1e-6


def test_sdf_alpha(golden):
    g = golden("march_r32")
    ridx = g["ray_indices"]
    dists = g["t_ends"] - g["t_starts"]
    for ca in (0.0, 0.5, 1.0):
        alpha, grad, feat, inv_s, sdf, nh = om.sdf_alpha(g.sd, g["sample_pts"], g["sample_levels"], dists,
                                                         g["dirs"][ridx], ca, AABB, GS, 3)
        assert rel_err(alpha, g[f"alpha_{ca}"]) < 1e-5
    assert rel_err(grad, g["sa_grad"]) < 1e-5 and rel_err(feat, g["sa_feat"]) < 1e-5
    assert rel_err(sdf, g["sa_sdf"]) < 1e-6 and rel_err(nh, g["sa_hess"]) < 1e-4
    assert rel_err(inv_s, g["sa_inv_s"]) < 1e-6


def test_shape_shading(golden):
    g = golden("march_r32")
    ridx = g["ray_indices"]
    nrm = torch.nn.functional.normalize(g["sa_grad"], dim=-1)
    col, occ, rough, refl = om.shape_shade(g.sd, _env(g), g["fg_lut"], g["sample_pts"], nrm, -g["dirs"][ridx], g["sa_feat"])
    assert rel_err(col, g["shade_color"]) < 2e-5
    assert rel_err(occ, g["shade_occ_prob"]) < 1e-5
    assert rel_err(rough, g["shade_roughness"]) < 1e-6
    assert rel_err(refl, g["shade_reflective"]) < 1e-6


def test_render_core(golden):
    g = golden("march_r32")
    out = om.render_core(g.sd, _env(g), g["fg_lut"], g["rays_o"], g["dirs"], g["radiis"], g["rays_cos"],
                         g["t_starts"], g["t_ends"], g["ray_indices"], AABB, GS, 3, float(g["base_radii"]), 0.5)
    for k in ("ray_rgb", "acc", "normal", "gradient_error", "std", "loss_sparse"):
        assert rel_err(out[k], g["rc/" + k]) < 2e-5, k
    assert rel_err(out["loss_hessian"], g["rc/loss_hessian"]) < 1e-4


def test_render_core_validation_branch(golden):
    """oracle/march.py:render_core_validation against the imported reference's render_core(is_train=False) (golden march_eval_r32:
    bumpy field, sharp surface, smooth pre-filtered maps): every validation key, incl. the traced occlusion."""
    g, ge = golden("march_r32"), golden("march_eval_r32")
    sd = dict(g.sd)
    sd.update(ge.sd)
    env = {"specular": [ge[f"env_spec{i}"] for i in range(3)], "diffuse": ge["env_diffuse"]}
    val = om.render_core_validation(sd, env, g["fg_lut"], ge["rays_o"], ge["dirs"], ge["radiis"], ge["rays_cos"], ge["t_starts"], ge["t_ends"],
                                    ge["ray_indices"], AABB, GS, 3, float(g["base_radii"]), 1.0)
    assert int((ge["val/occ_prob_gt"] > 1e-3).sum()) > 40
    for k, v in val.items():
        ref = ge["val/" + k]
        assert rel_err(v.reshape(ref.shape), ref) < 2e-5, (k, rel_err(v.reshape(ref.shape), ref))
    assert {"depth", "occ_prob_gt", "normal_vis", "specular_direct_light", "indirect_light", "occ_prob", "albedo"} <= set(val)


def test_refine_hits(golden):
    """SDF refinement of mesh hits vs MaterialRenderer.trace_sdf_with_mesh run on the imported reference."""
    from oracle import refine, shading as osh
    g = golden("refine_r32")
    tr = osh.MeshTracer(g["verts"][g["faces"].long()])
    inters, normals, depth, hit = refine.refine_hits(g.sd, tr, g["rays_o"], g["rays_d"], AABB, GS, 3, float(g["inv_s"]),
                                                     float(g["unit_size"]))
    assert torch.equal(hit, g["hit"].bool()) and 0.05 < hit.float().mean() < 0.95
    assert rel_err(depth, g["depth"]) < 1e-5 and rel_err(inters, g["inters"]) < 1e-5
    assert rel_err(normals, g["normals"]) < 1e-4


def test_material_nvs_frame(golden):
    """The oracle's MaterialRenderer.nvs composition (ray constructor, SDF-refined mesh hits, both shading passes, the frame's
    conventions) against a 24 x 24 frame rendered by the imported reference (golden material_nvs_r32; materialRenderer.py:641-752)."""
    from oracle import refine, shading as osh
    g, geo, small = golden("material_nvs_r32"), golden("refine_r32"), golden("shading_small")
    tr = osh.MeshTracer(g["verts"][g["faces"].long()])
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in small["sn"]]
    h, w = [int(v) for v in g["nvs_hw"]]
    frame, (inters, normals, depth, hit) = refine.material_nvs(small.sd, geo.sd, tr, g["nvs_pose"], g["nvs_K"], h, w, AABB, GS, 3,
                                                               float(geo["inv_s"]), float(g["unit_size"]), n_fd, (sn_d, sn_s), n_fs)
    assert torch.equal(hit, g["hit"].bool()) and 0.2 < hit.float().mean() < 0.8
    assert rel_err(inters[hit], g["inters"][hit]) < 1e-5 and rel_err(normals[hit], g["normals"][hit]) < 1e-4
    nvs = {k[4:]: v for k, v in g.a.items() if k.startswith("nvs/")}
    assert set(frame) == set(nvs) and len(nvs) == 15
    for k, ref in nvs.items():
        assert frame[k].shape == ref.shape, k
        assert rel_err(frame[k], ref) < 5e-5, (k, rel_err(frame[k], ref))
    assert float(nvs["occ_trace"][hit.reshape(h, w)].min()) < 0.9                  # the ring is seen by the secondary rays
    miss = ~hit
    assert (nvs["color"].reshape(-1, 3)[miss] == 1).all()
    # the reference assigns the (0,0,1) normal of a missing pixel inside `if sum(hit) > 0` of its 512-ray chunk loop: the frame's last
    # 64 rays (a chunk without a hit) keep zeros
    assert (nvs["normal"].reshape(-1, 3)[:512][miss[:512]] == torch.tensor([0.0, 0, 1])).all() and not hit[512:].any()
    assert float(nvs["normal"].reshape(-1, 3)[512:].abs().max()) == 0.0
    assert float(nvs["variance_diffuse_vis"].abs().max()) == 0.0


def test_oracle_occupancy_cell_marcher_and_update():
    """The oracle's stand-in for nerfacc's occupancy grid (third-party, absent: parity unpinned): cell lookup, stratified start, EMA
    update rule -- properties the build's kernels are then held to bit for bit (tests/test_gpu_renderers.py)."""
    from oracle import march as om
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    vol = torch.zeros(4, 4, 4, dtype=torch.uint8)
    vol[3, 0, 2] = 1                                   # cell x in [0.5,1], y in [-1,-0.5], z in [0,0.5]
    o = torch.tensor([[-2.0, -0.75, 0.25]])
    d = torch.tensor([[1.0, 0.0, 0.0]])
    t0, t1, ridx = om.march_uniform(o, d, torch.zeros(1), torch.full((1,), 10.0), aabb, 400, 0.01, vol, aabb, cells=True)
    mid = (t0 + t1) * 0.5 - 2.0
    assert ridx.numel() in (49, 50, 51) and float(mid.min()) >= 0.5 - 1e-6 and float(mid.max()) <= 1.0 + 1e-6
    tj0, _, _ = om.march_uniform(o, d, torch.zeros(1), torch.full((1,), 10.0), aabb, 400, 0.01, vol, aabb, cells=True,
                                 t_jitter=torch.tensor([0.004]))
    assert abs(float(tj0[0] - t0[0]) - 0.004) < 1e-6 or abs(float(tj0[0] - t0[0]) + 0.006) < 1e-6     # the lattice shifts by the jitter
    occs = torch.zeros(8)
    occs, b = om.occ_grid_update(occs, (1, 2, 2, 2), torch.tensor([1, 5]), torch.tensor([0.5, 0.001]))
    assert occs.tolist() == [0, 0.5, 0, 0, 0, 0.0010000000474974513, 0, 0] and b.reshape(-1).tolist() == [False, True] + [False] * 6
    occs, b = om.occ_grid_update(occs, (1, 2, 2, 2), torch.tensor([1]), torch.tensor([0.0]))
    assert abs(float(occs[1]) - 0.475) < 1e-7           # decays by 0.95 when the new evaluation is lower
