"""tensoflow_amd: MI355X-native (gfx950) implementation of TensoFlow's hot path.

The volumetric ray-march over the VM-decomposed tensorial SDF field and the flow-sampled
Monte-Carlo rendering integral, as hand-written HIP kernels behind a C-ABI shared library
(`libtensoflow_hip.so`, declared in include/tensoflow_hip.h), wrapped by modules that keep
the reference's `network.fields` / `network.flow` / `network.light` /
`network.shapeRenderer` / `network.materialRenderer` call surface.
"""
__version__ = "0.1.0"
