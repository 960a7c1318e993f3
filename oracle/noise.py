"""Counter-based noise stream shared by the oracle, the golden generator and the HIP parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference draws its randomness from
torch's global generators (`torch.randn`, `torch.randn_like`, `torch.rand`, `torch.randperm`,
`torch.multinomial`, `Normal.rsample`, `nn.Dropout`, `np.random.randint`); those streams cannot be
reproduced on a GPU.  SURVEY.md Appendix C.6 therefore replays every stochastic call from ONE
counter-based generator that exists on both sides: `numpy.random.Philox(key)`.  Draws are consumed in
the call order of the reference (documented next to each oracle function) so a golden fixture only
has to store the key and the expected outputs.

Every draw is produced in float64 by numpy and cast to float32, so the three consumers (patched
reference, oracle, HIP path in parity mode) see bit-identical noise tensors.
"""
from __future__ import annotations

import numpy as np


class NoiseStream:
    """Philox-keyed stream of the six kinds of draw the hot path consumes."""

    def __init__(self, key: int):
        self.key = int(key)
        self.gen = np.random.Generator(np.random.Philox(key=self.key))
        self.n_draws = 0
        self.log = []  # (kind, shape) in call order; used by tests to assert the documented order

    def _note(self, kind, shape):
        self.n_draws += 1
        self.log.append((kind, tuple(int(s) for s in shape)))

    # -- normal / uniform ---------------------------------------------------------------------
    def randn(self, *shape) -> np.ndarray:
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        self._note("randn", shape)
        return self.gen.standard_normal(size=shape).astype(np.float32)

    def rand(self, *shape) -> np.ndarray:
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        self._note("rand", shape)
        return self.gen.random(size=shape).astype(np.float32)

    # -- integer draws ------------------------------------------------------------------------
    def randperm(self, n: int) -> np.ndarray:
        self._note("randperm", (n,))
        return self.gen.permutation(int(n)).astype(np.int64)

    def randint(self, low: int, high: int, n: int) -> np.ndarray:
        self._note("randint", (n,))
        return self.gen.integers(int(low), int(high), size=int(n)).astype(np.int64)

    def multinomial(self, probs: np.ndarray, n: int) -> np.ndarray:
        """Inverse-CDF categorical sampling with replacement (stand-in for torch.multinomial)."""
        self._note("multinomial", (n,))
        u = self.gen.random(size=int(n))
        cdf = np.cumsum(np.asarray(probs, dtype=np.float64))
        cdf /= cdf[-1]
        return np.minimum(np.searchsorted(cdf, u, side="right"), len(cdf) - 1).astype(np.int64)

    # -- dropout ------------------------------------------------------------------------------
    def dropout_mask(self, shape, p: float) -> np.ndarray:
        """Scaled keep-mask: keep/(1-p) with P(keep)=1-p, float32 (what `x * mask` needs)."""
        shape = tuple(int(s) for s in shape)
        self._note("dropout", shape)
        keep = self.gen.random(size=shape) >= p
        return (keep.astype(np.float32) / np.float32(1.0 - p)).astype(np.float32)


class NullNoise:
    """Eval-mode stand-in: dropout masks are all-ones; any other draw is an error."""

    def dropout_mask(self, shape, p):
        return None

    def __getattr__(self, name):
        raise RuntimeError(f"NullNoise cannot serve '{name}' draws")


class TorchNoise:
    """Fast host RNG with the NoiseStream interface (torch's multi-threaded CPU generators).  Used ONLY by the
    cpu_baseline leg of bench.py, where the draw VALUES are irrelevant and numpy's single-threaded Philox would dominate
    the timing (a cfg-2 train_step consumes ~3e9 draws, mostly MINE dropout masks)."""

    def __init__(self, seed: int = 0):
        import torch
        self.t = torch
        self.g = torch.Generator().manual_seed(seed)

    def randn(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return self.t.randn(*shape, generator=self.g)

    def rand(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return self.t.rand(*shape, generator=self.g)

    def randperm(self, n):
        return self.t.randperm(int(n), generator=self.g)

    def randint(self, low, high, n):
        return self.t.randint(int(low), int(high), (int(n),), generator=self.g).numpy()

    def multinomial(self, probs, n):
        return self.t.multinomial(self.t.as_tensor(probs, dtype=self.t.float32), int(n), replacement=True, generator=self.g)

    def dropout_mask(self, shape, p):
        return (self.t.rand(*shape, generator=self.g) >= p).float() / (1.0 - p)
