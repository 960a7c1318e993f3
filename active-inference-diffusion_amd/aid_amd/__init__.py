"""aid_amd — MI355X-native (gfx950) implementation of the diffusion-policy training hot path of
neuronphysics/active-inference-diffusion, behind the reference's own Python surface.

Arithmetic lives in hand-written HIP kernels (csrc/, C ABI in include/aid_hip.h) reached through
`aid_amd._native`; PyTorch supplies device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
