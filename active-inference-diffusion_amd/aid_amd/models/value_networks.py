"""ValueNetwork V(s, t) (active_inference_diffusion/models/value_networks.py:9-60); forward on HIP kernels."""
import torch
import torch.nn as nn

from .. import _native as nat
from ._holders import TableCache, lin_ln_relu_stack, params_dict
from .score_networks import SinusoidalPositionEmbeddings


class ValueNetwork(nn.Module):
    def __init__(self, state_dim: int, hidden_dim: int = 256, time_embed_dim: int = 128, num_layers: int = 3):
        super().__init__()
        if num_layers != 3:
            raise NotImplementedError("the reference instantiates num_layers=3 (active_inference.py:93-98)")
        self.time_embed = nn.Sequential(SinusoidalPositionEmbeddings(time_embed_dim), nn.Linear(time_embed_dim, time_embed_dim), nn.ReLU())
        self.network = lin_ln_relu_stack([state_dim + time_embed_dim, hidden_dim, hidden_dim, hidden_dim], with_final=1)
        self._tab = TableCache(lambda: nat.value_table(params_dict(self), ""))
        self._gtab = TableCache(lambda: nat.value_table(params_dict(self, True), ""))
        self._ws = None

    def table(self):
        return self._tab.get(self)

    def grad_table(self):
        return self._gtab.get(self)

    def forward(self, state: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        if self._ws is None:
            self._ws = nat.Workspace(state.device)
        return nat.value_forward(self.table(), state.contiguous().float(), time.contiguous().float(), self._ws)
