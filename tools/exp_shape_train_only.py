"""Dev: the shape-stage training step alone (for rocprofv3 --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(bench.shape_train_probe(torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 5))
