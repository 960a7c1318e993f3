"""Where the hot path gets its randomness.

Production: `PhiloxNoise` hands every native op a (seed, offset) pair and the kernels draw on device
(counter-based Philox4x32-10, csrc/rng.hip).  Parity tests subclass `NoiseSource` and return explicit
tensors (the reference's draws replayed from oracle.noise.NoiseStream) — the product never imports the oracle.
Each method returns either None (=> device Philox) or the dict of tensors named in include/aid_hip.h.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


class NoiseSource:
    def __init__(self, seed: int = 0):
        self.seed = int(seed)
        self._offset = 0

    def next_offset(self, span: int = 1 << 32) -> int:
        """A fresh, non-overlapping Philox counter range for one native call."""
        o = self._offset
        self._offset += span
        return o

    # -- hooks (None => on-device Philox) ----------------------------------------------------------
    def replay_indices(self, size: int, batch: int) -> Optional[np.ndarray]:
        return None

    def belief(self, B: int, L: int, D: int, N: int, train: bool) -> Optional[dict]:
        return None

    def elbo(self, B: int, L: int, D: int, first_call: bool, time_importance) -> Optional[dict]:
        return None

    def efe(self, B: int, L: int, A: int, H: int, K: int, S: int) -> Optional[dict]:
        return None


class PhiloxNoise(NoiseSource):
    """Default: everything drawn on device."""
