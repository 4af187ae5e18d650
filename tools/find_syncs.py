"""Dev tool: every host synchronisation of one training step of either stage (torch.cuda.set_sync_debug_mode("warn")), with the
tensoflow_amd source lines that led to it.  A blocking copy at the top of a step keeps it from being queued under the previous
step's backward pass.
    python tools/find_syncs.py [mat|shape]"""
import os
import sys
import traceback
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    frames = [f for f in traceback.extract_stack() if "/tensoflow_amd/" in f.filename or "/bench.py" in f.filename]
    print("SYNC:", str(message)[:60], "<-", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-5:][::-1]))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "mat"
    dev = torch.device("cuda:0")
    if which == "mat":
        from tensoflow_amd.network.fields import MCShadingNetwork
        from tensoflow_amd.synth import sphere_surface_points, sphere_torus_mesh
        verts, faces = sphere_torus_mesh(224, 448, 256, 128)
        aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        torch.manual_seed(6033)
        m = MCShadingNetwork({"nis_diffuse_sample_num": 128, "nis_specular_sample_num": 128, "outer_light_version": "envlight"}, (verts, faces), aabb, 2.0 / 511)
        for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
            for p in fl.parameters():
                p.requires_grad = False
        m.train()
        m.use_flow_diffuse_copy = m.use_flow_specular_copy = True
        pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(2048, seed=99)]
        w = torch.rand(2048, 3, device=dev)

        def step():
            m.zero_grad(set_to_none=True)
            colors, out = m(pts, view, nrm, None, 600, True)
            ((colors * w).sum() + out["loss_nis"]).backward()
    else:
        from tensoflow_amd.network.shapeRenderer import ShapeRenderer
        from tensoflow_amd.synth import pinhole_rays, random_sdf_state, random_shape_shader_state
        R, n_rays = 300, 1024
        cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cuda",
                   nerfDataType=True, clip_sample_variance=False, apply_occ_loss=False)
        r = ShapeRenderer(cfg, training=False)
        sd = {"sdf_network." + k: v for k, v in random_sdf_state(seed=1, R=R).items()}
        sd.update(random_shape_shader_state(seed=8))
        r.load_state_dict(sd, strict=False)
        r.train()
        o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(n_rays, seed=2)]
        near, far = r.near_far_from_sphere(o, d)
        batch = {"rays_o": o, "rays_d": d, "dirs": d, "radiis": radii, "rays_cos": cos}
        target = torch.rand(n_rays, 3, device=dev)

        def step():
            r.zero_grad(set_to_none=True)
            r.color_network.envlight.build_mips()
            out = r.render(batch, near, far, None, perturb_overwrite=0, cos_anneal_ratio=0.5, is_train=True, step=2000)
            loss = ((out["ray_rgb"] - target) ** 2).mean() + 0.1 * out["gradient_error"].mean() + 0.1 * out["loss_sparse"] \
                + 5e-4 * out["loss_hessian"] + out["loss_tv_sdf"]
            loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    warnings.simplefilter("always")
    warnings.showwarning = show
    torch.cuda.set_sync_debug_mode("warn")
    step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print("done", which)


if __name__ == "__main__":
    main()
