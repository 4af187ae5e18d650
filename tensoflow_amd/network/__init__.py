"""Drop-in mirrors of the reference's `network` package for the hot path.

Same class names, constructor arguments, method names and `state_dict` keys as network/flow.py, network/fields.py,
network/light.py, network/shapeRenderer.py and network/materialRenderer.py, so a TrainerInv checkpoint loads unchanged
and `from network.shapeRenderer import ShapeRenderer` can become
`from tensoflow_amd.network.shapeRenderer import ShapeRenderer`.

The arithmetic runs in libtensoflow_hip.so (no PyTorch fallback).  Without autograd every module takes its fused
inference kernels; with autograd enabled on trainable parameters the renderers and MCShadingNetwork switch to the autograd
ops of tensoflow_amd.autograd (HIP forward + HIP / library-GEMM backward).  The stand-alone fused forwards of TensoSDF and
TensoFlow.sample raise when called with autograd on trainable parameters instead of silently detaching.  The dataset side of
the renderers (image tables, ray shuffling, train_step / test_step) is outside the hot path.
"""
