"""Dev: the material-stage training step alone (for rocprofv3 --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
from tensoflow_amd.synth import sphere_torus_mesh
verts, faces = sphere_torus_mesh(224, 448, 256, 128)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
print(bench.train_probe(dev, verts, faces, aabb, 2.0 / 511, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 10))
