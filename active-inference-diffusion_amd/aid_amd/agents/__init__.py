from .base_agent import BaseActiveInferenceAgent, RunningMeanStd
from .pixel_agent import DiffusionPixelAgent
from .state_agent import DiffusionStateAgent

# README spelling of the reference (README.md:33,51)
StateBasedAgent = DiffusionStateAgent
PixelBasedAgent = DiffusionPixelAgent

__all__ = ["BaseActiveInferenceAgent", "DiffusionStateAgent", "StateBasedAgent", "DiffusionPixelAgent", "PixelBasedAgent", "RunningMeanStd"]
