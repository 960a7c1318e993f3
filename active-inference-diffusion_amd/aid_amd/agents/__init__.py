from .base_agent import BaseActiveInferenceAgent, RunningMeanStd
from .state_agent import DiffusionStateAgent

# README spelling of the reference (README.md:33,51)
StateBasedAgent = DiffusionStateAgent

__all__ = ["BaseActiveInferenceAgent", "DiffusionStateAgent", "StateBasedAgent", "RunningMeanStd"]
