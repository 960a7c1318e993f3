"""LatentDynamicsModel f(s, a) -> s' (active_inference_diffusion/models/dynamics_models.py:9-68).
The HIP entry point computes `2 z + MLP([z|a])` = predict_next_latent (the reference adds the latent twice,
SURVEY D7); this module's own forward returns the model output `z + MLP` by subtracting one residual."""
import torch
import torch.nn as nn

from .. import _native as nat
from ._holders import TableCache, lin_ln_relu_stack, params_dict


class LatentDynamicsModel(nn.Module):
    def __init__(self, state_dim: int, action_dim: int, hidden_dim: int = 256, num_layers: int = 3, residual: bool = True):
        super().__init__()
        if num_layers != 3 or not residual:
            raise NotImplementedError("the reference instantiates num_layers=3, residual=True (active_inference.py:102-107)")
        self.state_dim, self.action_dim, self.residual = state_dim, action_dim, residual
        self.network = lin_ln_relu_stack([state_dim + action_dim, hidden_dim, hidden_dim, hidden_dim], with_final=state_dim)
        nn.init.uniform_(self.network[-1].weight, -1e-3, 1e-3)
        nn.init.zeros_(self.network[-1].bias)
        self._tab = TableCache(lambda: nat.dynamics_table(params_dict(self), action_dim, ""))
        self._gtab = TableCache(lambda: nat.dynamics_table(params_dict(self, True), action_dim, ""))
        self._ws = None

    def table(self):
        return self._tab.get(self)

    def grad_table(self):
        return self._gtab.get(self)

    def predict_next_mean(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        if self._ws is None:
            self._ws = nat.Workspace(state.device)
        return nat.predict_next_latent(self.table(), state.contiguous().float(), action.contiguous().float(), self._ws)

    def forward(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        return self.predict_next_mean(state, action) - state
