"""Dev tool: the material-stage training probe of bench.py alone (ms per step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    from tensoflow_amd.synth import sphere_torus_mesh
    verts, faces = sphere_torus_mesh(224, 448, 256, 128)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    print(bench.train_probe(torch.device("cuda:0"), verts, faces, aabb, 2.0 / 511, 128, steps))
