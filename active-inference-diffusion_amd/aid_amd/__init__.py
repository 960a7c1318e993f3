"""aid_amd — MI355X-native (gfx950) implementation of the diffusion-policy training hot path of
neuronphysics/active-inference-diffusion, behind the reference's own Python surface.

Arithmetic lives in hand-written HIP kernels (csrc/, C ABI in include/aid_hip.h) reached through
`aid_amd._native`; PyTorch supplies device memory, streams and torch.distributed only.  There is no CPU path.
"""
from .configs.config import (ActiveInferenceConfig, BeliefDynamicsConfig, DiffusionConfig, PixelObservationConfig, TrainingConfig)

__version__ = "0.1.0"
__all__ = ["ActiveInferenceConfig", "BeliefDynamicsConfig", "DiffusionConfig", "PixelObservationConfig", "TrainingConfig",
           "DiffusionStateAgent", "StateBasedAgent", "DiffusionPixelAgent", "PixelBasedAgent", "DrQV2Encoder",
           "RandomShiftAugmentation", "DiffusionActiveInference", "LatentDiffusionProcess", "BeliefDynamics", "FreeEnergyComputation", "set_float32_matmul_emulation"]


def set_float32_matmul_emulation(mode: str = "off"):
    """Arithmetic of the tall full-row GEMMs (the epistemic estimator's 128 000 / 640 000-row layers, the reverse loop's hoisted conditioning):
    "off" (default): `v_mfma_f32_32x32x2_f32`; "bf16x6": every float32 operand split exactly into three bfloat16 values, the six leading partial products
    accumulated in float32 on `v_mfma_f32_32x32x16_bf16` - measured closer to a float64 product than the float32 MFMA chain, at 2.67 x its rate per clock
    (csrc/gemm_rowln_x6.hip).  Process-wide; the environment variable AID_GEMM_X6=1 sets the initial value."""
    from . import _native
    if mode not in ("off", "bf16x6"):
        raise ValueError("set_float32_matmul_emulation: 'off' or 'bf16x6'")
    _native.gemm_emulation(1 if mode == "bf16x6" else 0)


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
