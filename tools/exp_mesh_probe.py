"""Dev: which synthetic SDF states give a usable iso-surface for tests/test_mesh.py."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from tensoflow_amd.mesh import extract_mesh
from tensoflow_amd.network.shapeRenderer import ShapeRenderer
from tensoflow_amd.synth import random_sdf_state
for name, R, load in (("default-init", 64, False), ("random R=64", 64, True), ("random R=128", 128, True)):
    torch.manual_seed(0)
    r = ShapeRenderer(dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cuda",
                           nerfDataType=True, blend_ratio=0.2), training=False)
    if load:
        r.load_state_dict({"sdf_network." + k: v for k, v in random_sdf_state(seed=1, R=R).items()}, strict=False)
    v, f = extract_mesh(r, resolution=96)
    rad = np.linalg.norm(v, axis=1) if len(v) else np.zeros(1)
    print(name, "tris", f.shape[0], "radius", rad.min(), rad.max(), "extent", v.min(0) if len(v) else None, v.max(0) if len(v) else None)
