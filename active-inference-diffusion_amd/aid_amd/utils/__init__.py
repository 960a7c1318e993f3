from .buffers import ReplayBuffer

__all__ = ["ReplayBuffer"]
