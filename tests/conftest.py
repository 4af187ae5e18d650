import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """npz fixture: arrays as torch tensors, `sd/...` entries collected into .sd."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.sd, self.grad, self.out, self.a = {}, {}, {}, {}
        for k in z.files:
            t = torch.from_numpy(z[k])
            if k.startswith("sd/"):
                self.sd[k[3:]] = t
            elif k.startswith("grad/"):
                self.grad[k[5:]] = t
            elif k.startswith("out/"):
                self.out[k[4:]] = t
            else:
                self.a[k] = t

    def __getitem__(self, k):
        return self.a[k]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def rel_err(a, b):
    """max |a-b| / max(|b|, 1) -- the 'relative fp32' measure used throughout.  For quantities below 1 (colours, alpha, sdf,
    roughness, weights) this is an ABSOLUTE bound; `true_rel_err` below is the relative one."""
    a, b = a.double(), b.double()
    v = float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) if a.numel() else 0.0
    if os.environ.get("TF_PARITY_LOG"):      # dev: every call site with both measures (which sites can be held to the relative one?)
        import inspect
        fr = inspect.stack()[1]
        with open(os.environ["TF_PARITY_LOG"], "a") as f:
            f.write(f"{os.path.basename(fr.filename)}:{fr.lineno}\t{fr.function}\t{v:.3e}\t{true_rel_err(a, b):.3e}\t{float(b.abs().max()) if b.numel() else 0:.3e}\n")
    return v


def true_rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor * max|b|): a genuinely relative measure with a small floor against division by ~0 (a quantity
    of size 1e-2 held to 1e-4 here is held to 1e-6 absolute)."""
    a, b = a.double(), b.double()
    if not a.numel():
        return 0.0
    return float(((a - b).abs() / b.abs().clamp_min(floor * float(b.abs().max()) + 1e-300)).max())


AABB = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
