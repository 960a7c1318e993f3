"""ReplayBuffer (active_inference_diffusion/utils/buffers.py:12-141) with the ring arrays resident in HBM.

Same constructor and add/sample/__len__ API.  `sample` draws indices exactly like the reference
(np.random.randint on the host — or from an injected NoiseSource), uploads the B int64 indices and gathers the
minibatch with one coalesced HIP kernel (aid_replay_gather): the 5 host->device minibatch copies of the
reference's train_step (state_agent.py:118-122) disappear.  The lz4-compressed branch of the reference is
unreachable and broken and is not reproduced.  `dones` come back as float 0/1 (what the kernels take).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from .. import _native as nat


class ReplayBuffer:
    def __init__(self, capacity: int, obs_shape: Tuple[int, ...], action_dim: int, device, optimize_memory: bool = False):
        self.capacity, self.device, self.obs_shape = int(capacity), torch.device(device), tuple(obs_shape)
        self.pos = self.size = 0
        self.version = 0            # bumped by every insertion (train_step's cross-step prefetch checks it)
        self.compress, self.dtype, self.optimize_memory = False, np.float32, bool(optimize_memory)
        obs_dim = int(np.prod(obs_shape))
        self._obs_dim, self.action_dim = obs_dim, action_dim
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.observations, self.next_observations = z(capacity, obs_dim), z(capacity, obs_dim)
        self.actions, self.rewards, self.dones = z(capacity, action_dim), z(capacity), z(capacity)
        self.index_source = None   # optional NoiseSource for parity tests

    def _maybe_switch_to_bytes(self, obs):
        """optimize_memory=True and uint8 frames arriving into an empty buffer: keep them as bytes in HBM (4x smaller ring arrays; the
        gather kernel widens them to the same float values the reference's float32 arrays would hold).  The reference's own
        `optimize_memory` branch (lz4) is broken."""
        if self.optimize_memory and self.size == 0 and self.observations.dtype != torch.uint8 and np.asarray(obs).dtype == np.uint8:
            self.observations = torch.zeros(self.capacity, self._obs_dim, dtype=torch.uint8, device=self.device)
            self.next_observations = torch.zeros_like(self.observations)
            self.dtype = np.uint8

    def add(self, obs: np.ndarray, action: np.ndarray, reward: float, next_obs: np.ndarray, done: bool):
        i = self.pos
        self._maybe_switch_to_bytes(obs)
        if self.observations.dtype == torch.uint8:
            o, n = np.asarray(obs), np.asarray(next_obs)
            if o.dtype != np.uint8 or n.dtype != np.uint8:
                raise TypeError("this ReplayBuffer holds uint8 frames (first insertion was uint8): pass uint8 observations")
            self.observations[i] = torch.as_tensor(o.reshape(-1)); self.next_observations[i] = torch.as_tensor(n.reshape(-1))
        else:
            self.observations[i] = torch.as_tensor(np.asarray(obs, dtype=np.float32).reshape(-1))
            self.next_observations[i] = torch.as_tensor(np.asarray(next_obs, dtype=np.float32).reshape(-1))
        self.actions[i] = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(-1))
        self.rewards[i] = float(reward)
        self.dones[i] = float(bool(done))
        self.pos = (self.pos + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)
        self.version += 1

    def add_batch(self, obs, actions, rewards, next_obs, dones):
        """Bulk insert (synthetic-replay benches, vectorised collectors): one H->D copy per field."""
        n = len(rewards)
        assert n <= self.capacity
        idx = (torch.arange(n) + self.pos) % self.capacity
        f = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(n, -1).to(self.device)
        self._maybe_switch_to_bytes(obs)
        if self.observations.dtype == torch.uint8:
            u = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.uint8)).reshape(n, -1).to(self.device)
            self.observations[idx] = u(obs); self.next_observations[idx] = u(next_obs)
        else:
            self.observations[idx] = f(obs); self.next_observations[idx] = f(next_obs)
        self.actions[idx] = f(actions)
        self.rewards[idx] = f(rewards).reshape(n); self.dones[idx] = f(dones).reshape(n)
        self.pos = (self.pos + n) % self.capacity
        self.size = min(self.size + n, self.capacity)
        self.version += 1

    def sample(self, batch_size: int) -> Dict[str, torch.Tensor]:
        idx = self.index_source.replay_indices(self.size, batch_size) if self.index_source is not None else None
        if idx is None:
            idx = np.random.randint(0, self.size, batch_size)
        idx_dev = torch.as_tensor(np.asarray(idx, dtype=np.int64)).to(self.device)
        o, n, a, r, d = nat.replay_gather(self.observations, self.next_observations, self.actions, self.rewards, self.dones, idx_dev)
        shp = (batch_size, *self.obs_shape)
        return {"observations": o.view(shp), "actions": a, "rewards": r, "next_observations": n.view(shp), "dones": d}

    def __len__(self):
        return self.size
