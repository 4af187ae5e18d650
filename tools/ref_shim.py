"""Development-only shim that makes the reference importable on a CPU-only container.

Runs ONLY in the build container (needs /root/reference); never imported by tests that
run on the GPU box, by bench.py or by the product package.  It copies nothing from the
reference: it installs stub modules for the third-party packages that are absent here,
routes the three third-party *arithmetic* entry points the hot path crosses to the
oracle's restatements (oracle/texture.py, oracle/segments.py, oracle/mesh.py), and rewrites
'cuda' device requests to CPU so the reference's hard-coded `.cuda()` calls work.
"""
import contextlib
import os
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

REF_ROOT = os.environ.get("TENSOFLOW_REFERENCE", "/root/reference")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)


def _is_cuda(dev):
    if isinstance(dev, str):
        return dev.startswith("cuda")
    if isinstance(dev, torch.device):
        return dev.type == "cuda"
    return False


class CudaToCpu(TorchFunctionMode):
    """Rewrite device='cuda' kwargs / positional device args to CPU."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        if any(_is_cuda(a) for a in args):
            args = tuple("cpu" if _is_cuda(a) else a for a in args)
        return func(*args, **kwargs)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__["__getattr__"] = lambda k, n=name: _missing_attr(n, k)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Missing:
    def __init__(self, what):
        self._what = what

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self._what} is a stub (third-party package absent)")

    def __getattr__(self, k):
        return _Missing(f"{self._what}.{k}")


def _missing_attr(mod, k):
    if k.startswith("__"):
        raise AttributeError(k)
    return _Missing(f"{mod}.{k}")


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    from oracle import texture as otex
    from oracle import segments as oseg
    from oracle import mesh as omesh

    dr_torch = _stub("nvdiffrast.torch", texture=otex.texture)
    _stub("nvdiffrast", torch=dr_torch)
    _stub("torch_scatter", segment_coo=oseg.segment_coo)
    _stub("nerfacc",
          render_weight_from_alpha=oseg.render_weight_from_alpha,
          accumulate_along_rays=oseg.accumulate_along_rays,
          OccGridEstimator=_Missing("nerfacc.OccGridEstimator"))
    rt = _stub("raytracing", RayTracer=omesh.BruteForceRayTracer)
    _stub("_raytracing")
    for name in ["cv2", "open3d", "mcubes", "plyfile", "h5py", "ghalton", "imageio",
                 "trimesh", "tensorboardX", "lpips", "kornia", "omegaconf"]:
        if name not in sys.modules:
            _stub(name, __getattr__=lambda k, n=name: _Missing(f"{n}.{k}"))
    sk = _stub("skimage")
    sk.io = _stub("skimage.io", imread=_Missing("skimage.io.imread"), imsave=_Missing("imsave"))
    sk.metrics = _stub("skimage.metrics", structural_similarity=_Missing("ssim"))
    sk.measure = _stub("skimage.measure")
    t3 = _stub("transforms3d")
    t3.axangles = _stub("transforms3d.axangles", mat2axangle=_Missing("mat2axangle"))
    t3.euler = _stub("transforms3d.euler", euler2mat=_Missing("euler2mat"))
    t3.quaternions = _stub("transforms3d.quaternions", mat2quat=_Missing("q"), quat2mat=_Missing("q"))
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", save_image=_Missing("save_image"), make_grid=_Missing("make_grid"))
    tv.transforms = _stub("torchvision.transforms")

    # no-op .cuda()
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if not hasattr(__import__("numpy"), "math"):
        import math
        import numpy as np
        np.math = math  # utils/ref_utils.py uses np.math.factorial (removed in numpy 2)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


@contextlib.contextmanager
def reference():
    """Context: reference importable, cwd=/root/reference, cuda->cpu rewriting active."""
    install()
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        with CudaToCpu():
            yield
    finally:
        os.chdir(cwd)
