"""Parameter-holder helpers.  torch.nn containers are used ONLY to own Parameters under the reference's
state_dict names and to reuse its initialisers; none of their forward() methods is ever called — all
arithmetic goes through aid_amd._native (HIP)."""
from __future__ import annotations

import torch
import torch.nn as nn


def native_only(name: str):
    def _forward(self, *a, **k):
        raise RuntimeError(f"{name} is a parameter holder: call the owning network's forward (HIP path) instead")
    return _forward


def params_dict(module: nn.Module, grads: bool = False):
    """name -> tensor (parameter data or its .grad view) for the ctypes table builders."""
    if grads:
        out = {}
        for n, p in module.named_parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p.data)
            out[n] = p.grad
        return out
    return {n: p.data for n, p in module.named_parameters()}


class TableCache:
    """Rebuilds a ctypes pointer table only when the underlying storage moved (e.g. after load/.to())."""

    def __init__(self, build):
        self._build, self._key, self._val = build, None, None

    def get(self, module: nn.Module):
        key = tuple((p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in module.parameters())
        if key != self._key:
            self._val, self._key = self._build(), key
        return self._val


def lin_ln_relu_stack(dims, with_final=None):
    """[Linear, LayerNorm, ReLU] * n (+ optional final Linear) as an nn.Sequential with reference indices."""
    layers = []
    for i in range(len(dims) - 1):
        layers += [nn.Linear(dims[i], dims[i + 1]), nn.LayerNorm(dims[i + 1]), nn.ReLU()]
    if with_final is not None:
        layers.append(nn.Linear(dims[-1], with_final))
    return nn.Sequential(*layers)
