"""Dev tool: the shape-stage training probe of bench.py alone (ms per step), optionally with the composed backward for an A/B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    print(bench.shape_train_probe(torch.device("cuda:0"), steps))
