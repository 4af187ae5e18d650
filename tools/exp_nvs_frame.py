"""Dev: full 800x800 frame through the drop-in ShapeRenderer.nvs (sample_ray 64 + 4x16 importance samples, render_core,
validation branch) at the BASELINE field size (R = 300, C = 36, 3 mips)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensoflow_amd.network.shapeRenderer import ShapeRenderer
from tensoflow_amd.synth import random_sdf_state, random_shape_shader_state
R = 300
cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cuda", nerfDataType=True,
           clip_sample_variance=False, test_ray_num=int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
r = ShapeRenderer(cfg, training=False)
sd = {"sdf_network." + k: v for k, v in random_sdf_state(seed=1, R=R).items()}
sd.update(random_shape_shader_state(seed=8))
missing, unexpected = r.load_state_dict(sd, strict=False)
print("missing", len(missing), "unexpected", len(unexpected))
r.eval()
r.color_network.envlight.build_mips()
c2w = np.array([[1.0, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 2.0]], np.float32)
K = np.array([[1111.1, 0, 400], [0, 1111.1, 400], [0, 0, 1]], np.float32)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = r.nvs(c2w, K, 800, 800)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"nvs 800x800: {dt*1e3:.0f} ms  {640000/dt/1e6:.2f} M rays/s  coverage {float((img['color'].min(-1) < 0.999).mean()):.3f}")
