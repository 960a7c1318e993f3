"""Configuration dataclasses — field-for-field the reference surface
(active_inference_diffusion/configs/config.py:10-126): same names, defaults and nesting, so code that
builds `ActiveInferenceConfig(...)` for the reference builds it unchanged for this package."""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch


@dataclass
class DiffusionConfig:
    num_diffusion_steps: int = 1000
    beta_start: float = 1e-4
    beta_end: float = 0.02
    beta_schedule: str = "cosine"  # "cosine" | "linear"
    prediction_type: str = "score"
    use_continuous_time: bool = True
    time_annealing_start: float = 1.0
    time_annealing_end: float = 0.1
    annealing_steps: int = 100000
    gradient_clip_val: float = 0.1


@dataclass
class BeliefDynamicsConfig:
    use_belief_dynamics: bool = True
    belief_dim: int = 50
    diffusion_coefficient: float = 0.1
    learning_rate: float = 0.1
    dt: float = 0.01
    min_variance: float = 1e-6
    max_variance: float = 10.0
    use_full_covariance: bool = False
    noise_scale: float = 0.01


@dataclass
class ActiveInferenceConfig:
    env_name: str = "HalfCheetah-v4"
    observation_dim: int = 17
    action_dim: int = 6
    precision_init: float = 1.0
    expected_free_energy_horizon: int = 5
    efe_horizon: int = 5
    epistemic_weight: float = 0.1
    extrinsic_weight: float = 1.0
    pragmatic_weight: float = 1.0
    consistency_weight: float = 0.1
    discount_factor: float = 0.99
    contrastive_weight: float = 0.5
    kl_weight: float = 0.1
    diffusion_weight: float = 1.0
    reward_weight: float = 0.5
    hidden_dim: int = 512
    latent_dim: int = 128
    spatial_aggregator_output_dim: int = 256
    num_layers: int = 3
    pixel_observation: bool = False
    batch_size: int = 256
    learning_rate: float = 5e-5
    gradient_clip: float = 0.5
    preference_temperature: float = 1.0
    preference_learning_rate: float = 0.01
    min_preference_temperature: float = 0.1
    max_preference_temperature: float = 10.0
    temperature_decay: float = 0.995
    use_reward_preferences: bool = True
    baseline_reward: float = 0.0
    preference_momentum: float = 0.9
    diffusion: DiffusionConfig = field(default_factory=DiffusionConfig)
    belief_dynamics: BeliefDynamicsConfig = field(default_factory=BeliefDynamicsConfig)
    device: str = "cuda" if torch.cuda.is_available() else "cpu"


@dataclass
class PixelObservationConfig:
    image_shape: Tuple[int, int, int] = (3, 84, 84)
    frame_stack: int = 3
    encoder_type: str = "drqv2"
    encoder_feature_dim: int = 80
    augmentation: bool = True
    random_shift_pad: int = 4
    pixel_observation: bool = True


@dataclass
class TrainingConfig:
    total_timesteps: int = 1_000_000
    eval_frequency: int = 5_000
    save_frequency: int = 50_000
    log_frequency: int = 1_000
    exploration_noise: float = 0.1
    exploration_decay: float = 0.999
    min_exploration: float = 0.01
    buffer_size: int = 100_000
    learning_starts: int = 5_000
    train_frequency: int = 2
    gradient_steps: int = 4
    num_parallel_envs: int = 6
    num_eval_episodes: int = 10
    use_wandb: bool = True
    project_name: str = "active-inference-diffusion"
    experiment_name: Optional[str] = None
