"""Drop-in mirrors of the reference's `network` package for the hot path.

Same class names, constructor arguments, method names and `state_dict` keys as network/flow.py, network/fields.py,
network/light.py, network/shapeRenderer.py and network/materialRenderer.py, so a TrainerInv checkpoint loads unchanged
and `from network.shapeRenderer import ShapeRenderer` can become
`from tensoflow_amd.network.shapeRenderer import ShapeRenderer`.

The arithmetic runs in libtensoflow_hip.so (a missing library raises; there is no CPU path).  Without autograd every module takes
its fused inference kernels; with autograd enabled on trainable parameters the renderers and MCShadingNetwork switch to the
autograd ops of tensoflow_amd.autograd: HIP forward and HIP backward for the field gathers, compositing, flow densities, BRDF
weights, cube maps AND every dense layer (tf_linear_fwd / tf_linear_bwd: the nn.Sequential modules only hold the parameters --
no library GEMM runs in a training step); elementwise algebra between the kernels is device-resident torch.  The stand-alone
fused forwards of TensoSDF and TensoFlow.sample raise when called with autograd on trainable parameters instead of silently
detaching.  The dataset side of the renderers reads TensoSDF synthetic scenes.
"""
