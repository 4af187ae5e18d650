"""Drop-in mirrors of the reference's `network` package for the hot path.

Same class names, constructor arguments, method names and `state_dict` keys as network/flow.py,
network/fields.py, network/light.py, so a TrainerInv checkpoint loads unchanged and
`from network.flow import TensoFlow` can become `from tensoflow_amd.network.flow import TensoFlow`.
The arithmetic runs in libtensoflow_hip.so (no PyTorch fallback).  Round 1 covers the forward
(inference / `torch.no_grad`) direction of every module; parameter gradients exist for the VM gather,
cube-map lookup and compositing ops (see ops.py) and arrive for the fused decoders in a later round --
calling a fused forward with autograd enabled on trainable parameters raises instead of silently
detaching.
"""
