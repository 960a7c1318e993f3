from .config import (ActiveInferenceConfig, BeliefDynamicsConfig, DiffusionConfig, PixelObservationConfig, TrainingConfig)

__all__ = ["DiffusionConfig", "BeliefDynamicsConfig", "ActiveInferenceConfig", "PixelObservationConfig", "TrainingConfig"]
