"""aid_amd — MI355X-native (gfx950) implementation of the diffusion-policy training hot path of
neuronphysics/active-inference-diffusion, behind the reference's own Python surface.

Arithmetic lives in hand-written HIP kernels (csrc/, C ABI in include/aid_hip.h) reached through
`aid_amd._native`; PyTorch supplies device memory, streams and torch.distributed only.  There is no CPU path.
"""
from .configs.config import (ActiveInferenceConfig, BeliefDynamicsConfig, DiffusionConfig, PixelObservationConfig, TrainingConfig)

__version__ = "0.1.0"
__all__ = ["ActiveInferenceConfig", "BeliefDynamicsConfig", "DiffusionConfig", "PixelObservationConfig", "TrainingConfig",
           "DiffusionStateAgent", "StateBasedAgent", "DiffusionPixelAgent", "PixelBasedAgent", "DrQV2Encoder",
           "RandomShiftAugmentation", "DiffusionActiveInference", "LatentDiffusionProcess", "BeliefDynamics", "FreeEnergyComputation"]


def __getattr__(name):  # lazy: importing the package must not require the GPU library (build() imports it on CPU)
    if name in ("DrQV2Encoder", "RandomShiftAugmentation", "SpatialAttention"):
        from . import encoder
        return getattr(encoder, name)
    if name in ("DiffusionStateAgent", "StateBasedAgent", "DiffusionPixelAgent", "PixelBasedAgent", "BaseActiveInferenceAgent"):
        from . import agents
        return getattr(agents, name)
    if name in ("DiffusionActiveInference", "LatentDiffusionProcess", "BeliefDynamics", "FreeEnergyComputation", "EMAModel"):
        from . import core
        return getattr(core, name)
    raise AttributeError(name)
