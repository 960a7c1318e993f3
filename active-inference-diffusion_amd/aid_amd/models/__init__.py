from .dynamics_models import LatentDynamicsModel
from .policy_networks import DiffusionConditionedPolicy
from .score_networks import LatentScoreNetwork, SinusoidalPositionEmbeddings
from .value_networks import ValueNetwork

__all__ = ["LatentScoreNetwork", "SinusoidalPositionEmbeddings", "DiffusionConditionedPolicy", "ValueNetwork", "LatentDynamicsModel"]
