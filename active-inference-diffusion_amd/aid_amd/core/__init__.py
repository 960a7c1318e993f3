from ..configs.config import (ActiveInferenceConfig, BeliefDynamicsConfig, DiffusionConfig, PixelObservationConfig, TrainingConfig)
from .active_inference import DiffusionActiveInference, EMAModel, FunctionSpaceEpistemicEstimator
from .belief_dynamics import BeliefDynamics
from .diffusion import LatentDiffusionProcess
from .free_energy import FreeEnergyComputation

__all__ = ["DiffusionActiveInference", "LatentDiffusionProcess", "BeliefDynamics", "FreeEnergyComputation", "ActiveInferenceConfig",
           "PixelObservationConfig", "TrainingConfig", "DiffusionConfig", "BeliefDynamicsConfig", "EMAModel"]
