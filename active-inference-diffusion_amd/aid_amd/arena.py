"""Flat parameter arenas: every optimiser group lives in ONE contiguous float32 buffer on the GPU.

Why (MI355X): clip_grad_norm_ -> AdamW -> EMA becomes a single HBM-streaming kernel over the arena
(`aid_adamw_step`, 36-40 B/param) instead of ~300 tiny per-tensor kernels, the data-parallel gradient exchange
is one RCCL all-reduce per group on a contiguous buffer (no bucketing copies), and the 13 adaLN modulation
matrices sit back to back so they run as one GEMM.  The nn.Parameters keep their reference names and shapes
(they are views into the arena), so state_dict()/load_state_dict() interchange with the reference.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn

from . import _native as nat
from . import parallel

ALIGN = 64  # floats (256 B): keeps every tensor 16-B aligned for vector loads and LDS-DMA


class ParamArena:
    def __init__(self, named_params: Sequence[Tuple[str, nn.Parameter]], device, with_ema: bool = False, trainable: bool = True):
        """trainable=False: parameters only (no gradient / moment buffers) - the never-stepped decoder, reward predictor, ConvDecoder."""
        self.device = torch.device(device)
        self.names: List[str] = [n for n, _ in named_params]
        self.params: List[nn.Parameter] = [p for _, p in named_params]
        self.offsets: Dict[str, Tuple[int, torch.Size]] = {}
        off = 0
        for n, p in named_params:
            self.offsets[n] = (off, p.shape)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.trainable = trainable
        self.grad = torch.zeros_like(self.flat) if trainable else self.flat.new_zeros(0)
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)
        self.ema = None
        self.step_count = 0
        self._scratch = torch.zeros(8192, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for n, p in named_params:
                o, shp = self.offsets[n]
                view = self.flat[o:o + p.numel()].view(shp)
                view.copy_(p.data.to(self.device, torch.float32))
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(shp) if trainable else None
        if with_ema:
            self.ema = self.flat.clone()

    # -- views ------------------------------------------------------------------------------------
    def grad_views(self) -> Dict[str, torch.Tensor]:
        return {n: self.grad[o:o + int(torch.Size(s).numel())].view(s) for n, (o, s) in self.offsets.items()}

    def ema_views(self) -> Dict[str, torch.Tensor]:
        assert self.ema is not None
        return {n: self.ema[o:o + int(torch.Size(s).numel())].view(s) for n, (o, s) in self.offsets.items()}

    def reset_ema(self):
        if self.ema is not None:
            self.ema.copy_(self.flat)

    def zero_grad(self):
        self.grad.zero_()

    def rebind(self):
        """Re-attach .grad views (torch optimizers / user code may have set them to None)."""
        for n, p in zip(self.names, self.params):
            o, shp = self.offsets[n]
            if p.data.data_ptr() != self.flat.data_ptr() + 4 * o:
                with torch.no_grad():
                    self.flat[o:o + p.numel()].view(shp).copy_(p.data)
                p.data = self.flat[o:o + p.numel()].view(shp)
            p.grad = self.grad[o:o + p.numel()].view(shp) if self.trainable else None

    # -- fused clip -> AdamW -> EMA -----------------------------------------------------------------
    def step(self, lr: float, weight_decay: float, clip_norm: float, ema_decay: float = 0.9999, ext_sumsq: torch.Tensor = None):
        """ext_sumsq: device scalar = sum of squared gradients of the OTHER arenas clipped together with this one."""
        self.step_count += 1
        if ext_sumsq is not None:
            nat.adamw_step_ext(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.ema, lr, weight_decay, self.step_count, clip_norm,
                               self._scratch, ext_sumsq, ema_decay=ema_decay)
        else:
            nat.adamw_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.ema, lr, weight_decay, self.step_count, clip_norm,
                           self._scratch, ema_decay=ema_decay)

    def grad_sumsq(self) -> torch.Tensor:
        """Device scalar (view of scratch[2]): sum of squares of this arena's gradients."""
        nat.grad_sumsq(self.grad, self._scratch)
        return self._scratch[2:3]

    def grad_norm(self) -> torch.Tensor:
        """Device scalar: the global L2 norm measured by the last step()."""
        return self._scratch[0]

    # -- optimiser state in torch.optim.AdamW's layout (checkpoint interchange) --------------------------
    def state_dict(self) -> dict:
        state = {}
        for i, n in enumerate(self.names):
            o, s = self.offsets[n]; k = int(torch.Size(s).numel())
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[o:o + k].view(s).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view(s).clone()}
        return {"state": state if self.step_count else {}, "param_groups": [{"params": list(range(len(self.names)))}]}

    def load_state_dict(self, sd: dict):
        for i, n in enumerate(self.names):
            st = sd.get("state", {}).get(i)
            if st is None:
                continue
            o, s = self.offsets[n]; k = int(torch.Size(s).numel())
            self.exp_avg[o:o + k].view(s).copy_(st["exp_avg"]); self.exp_avg_sq[o:o + k].view(s).copy_(st["exp_avg_sq"])
            self.step_count = int(float(st["step"]))


def adamw_state_dict(layout, hyper) -> dict:
    """torch.optim.AdamW.state_dict() for a parameter list laid out as `layout`: entry i is (arena, name) for a parameter this build steps,
    or None for one the reference passes to its optimiser but never updates (it keeps no state for those either).  One param group."""
    state = {}
    for i, e in enumerate(layout):
        if e is None or not e[0].step_count:
            continue
        a, n = e
        o, shp = a.offsets[n]; k = int(torch.Size(shp).numel())
        state[i] = {"step": torch.tensor(float(a.step_count)), "exp_avg": a.exp_avg[o:o + k].view(shp).clone(),
                    "exp_avg_sq": a.exp_avg_sq[o:o + k].view(shp).clone()}
    group = {"lr": hyper["lr"], "betas": tuple(hyper["betas"]), "eps": hyper["eps"], "weight_decay": hyper["weight_decay"], "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
             "params": list(range(len(layout)))}
    return {"state": state, "param_groups": [group]}


def adamw_load_state_dict(layout, hyper, sd: dict):
    arenas = []
    for i, e in enumerate(layout):
        st = sd.get("state", {}).get(i)
        if st is None or e is None:
            continue
        a, n = e
        o, shp = a.offsets[n]; k = int(torch.Size(shp).numel())
        a.exp_avg[o:o + k].view(shp).copy_(st["exp_avg"]); a.exp_avg_sq[o:o + k].view(shp).copy_(st["exp_avg_sq"])
        a.step_count = int(float(st["step"]))
        if a not in arenas:
            arenas.append(a)
    groups = sd.get("param_groups") or [{}]
    for k in ("lr", "weight_decay", "betas", "eps"):
        if k in groups[0]:
            hyper[k] = groups[0][k]


class ArenaOptimizer:
    """What callers see as `agent.score_optimizer` etc.: zero_grad / step / state_dict over one arena
    (reference: torch.optim.AdamW groups of aid/agents/state_agent.py:245-274)."""

    def __init__(self, arena: ParamArena, lr: float, weight_decay: float, clip_norm: float, ema_decay: float = 0.9999, layout=None):
        self.arena, self.lr, self.weight_decay, self.clip_norm, self.ema_decay = arena, lr, weight_decay, clip_norm, ema_decay
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": (0.9, 0.999), "eps": 1e-8}]
        # checkpoint layout = the parameter list the reference hands to torch.optim.AdamW (order matters: torch indexes state by position)
        self.layout = layout if layout is not None else [(arena, n) for n in arena.names]

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    _pending, _reduced = None, False

    def allreduce_grads(self, async_op: bool = False):
        """Data parallel: average this group's gradient arena over ranks (one RCCL all-reduce). No-op on one GPU.
        async_op: the collective is enqueued now and step() waits for it (on whatever stream step() is called from)."""
        self._pending = parallel.allreduce_mean_(self.arena.grad, async_op=async_op)
        self._reduced = True
        return self._pending

    def step(self, reduce: bool = True):
        if self._pending is not None:
            self._pending.wait()
        elif reduce and not self._reduced:
            parallel.allreduce_mean_(self.arena.grad)
        self._pending, self._reduced = None, False
        self.arena.step(self.param_groups[0]["lr"], self.param_groups[0]["weight_decay"], self.clip_norm, self.ema_decay)

    def state_dict(self):
        """torch.optim.AdamW's layout (per-parameter step / exp_avg / exp_avg_sq by position in the reference's parameter list; one param
        group with its hyper-parameters), so a checkpoint written by either side loads into the other (`utils/training.py:72-172` saves and
        restores `*_optimizer.state_dict()`)."""
        return adamw_state_dict(self.layout, self.param_groups[0])

    def load_state_dict(self, sd):
        adamw_load_state_dict(self.layout, self.param_groups[0], sd)
        self.lr, self.weight_decay = self.param_groups[0]["lr"], self.param_groups[0]["weight_decay"]


class AdamL2Optimizer:
    """torch.optim.Adam(weight_decay=...) semantics (L2 folded into the gradient) + clip_grad_norm_ over one arena: the estimator's
    optimiser of base_agent.py:134-139 (lr * 0.1, weight_decay 1e-5)."""

    def __init__(self, arena: ParamArena, lr: float, weight_decay: float, clip_norm: float):
        self.arena, self.clip_norm = arena, clip_norm
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": (0.9, 0.999), "eps": 1e-8}]

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    def step(self):
        parallel.allreduce_mean_(self.arena.grad)
        a, g = self.arena, self.param_groups[0]
        a.step_count += 1
        nat.adam_l2_step(a.flat, a.grad, a.exp_avg, a.exp_avg_sq, g["lr"], g["weight_decay"], a.step_count, self.clip_norm, a._scratch)

    def state_dict(self):
        return self.arena.state_dict()

    def load_state_dict(self, sd):
        self.arena.load_state_dict(sd)
