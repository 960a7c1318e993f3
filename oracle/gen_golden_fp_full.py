#!/usr/bin/env python3
"""Golden vectors for the full-covariance Fokker-Planck belief step, from the REFERENCE's own sub-functions (container-only).
`BeliefDynamics.update` raises before it returns (it differentiates a detached tensor), so the harness replays its body
(active_inference_diffusion/core/belief_dynamics.py:110-152) line by line with the reference's own `_compute_free_energy_gradient_autodiff`,
`_matrix_exponential_update`, `_ensure_numerical_stability`, `_safe_inverse`, `entropy` and `reset`, substituting only the Hessian the
broken autodiff call was meant to produce (that of the log-density the gradient comes from: -(1/ns^2 + 1) I).  The reference files are
not edited.  Asserts oracle == reference, then writes tests/golden/fp_full.npz."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aid_oracle as orc  # noqa: E402
from oracle import gen_golden as gg  # noqa: E402


def spd(rng, L, lo, hi):
    """Random symmetric positive-definite matrix with eigenvalues log-uniform in [lo, hi] (some below the clamp, wide condition number)."""
    Q, _ = np.linalg.qr(rng.standard_normal((L, L)))
    w = np.exp(rng.uniform(np.log(lo), np.log(hi), L))
    return (Q * w) @ Q.T


def main():
    gg.install_stubs()
    sys.path.insert(0, gg.REF)
    from active_inference_diffusion.core.belief_dynamics import BeliefDynamics
    from active_inference_diffusion.configs.config import BeliefDynamicsConfig
    rng = np.random.default_rng(2024)
    out = {}
    cases = {   # name: (L, config overrides, eigenvalue range of the initial covariance, steps)
        "sane": (17, dict(noise_scale=1.0, dt=0.05, diffusion_coefficient=0.3), (1e-3, 5.0), 3),
        "illcond": (24, dict(noise_scale=2.0, dt=0.01), (1e-10, 1e3), 2),          # eigenvalues below the clamp, condition number > 1e6
        "default": (6, dict(), (0.5, 2.0), 1),                                    # defaults: the factor is exp(400) per step
    }
    for name, (L, over, (lo, hi), steps) in cases.items():
        cfg = BeliefDynamicsConfig(use_full_covariance=True, **over)
        bd = BeliefDynamics(L, cfg)
        m0, c0 = rng.standard_normal(L), spd(rng, L, lo, hi)
        bd.reset(torch.from_numpy(m0), torch.from_numpy(c0))
        min_eig = max(cfg.min_variance, 1e-8)
        c_reset, cond0 = orc.fp_ensure_numerical_stability(c0, min_eig)
        np.testing.assert_allclose(bd.covariance.numpy(), c_reset, rtol=1e-9, atol=1e-12 * hi)
        np.testing.assert_allclose(bd.precision.numpy(), orc.fp_safe_inverse(c_reset, min_eig), rtol=1e-7)
        out[f"{name}_dims"] = np.array([L, steps]); out[f"{name}_cfg"] = np.array([cfg.dt, cfg.diffusion_coefficient, cfg.learning_rate, cfg.noise_scale, cfg.min_variance])
        out[f"{name}_mean_init"], out[f"{name}_cov_init"], out[f"{name}_cov_reset"] = m0, c0, bd.covariance.numpy().copy()
        mean_o, cov_o = m0.copy(), c_reset
        for t in range(steps):
            obs, sc, xi = rng.standard_normal(L), rng.standard_normal(L), rng.standard_normal(L)
            # ---- the body of BeliefDynamics.update (:110-152) with the reference's own functions
            dt, D, lr = cfg.dt, cfg.diffusion_coefficient, cfg.learning_rate
            o64, s64 = torch.from_numpy(obs), torch.from_numpy(sc)
            Fg = bd._compute_free_energy_gradient_autodiff(bd.mean, o64, s64, None)
            mean_noise = math.sqrt(2 * D * dt) * torch.from_numpy(xi) * cfg.noise_scale
            adaptive_dt = dt / (1 + 0.1 * Fg.norm())
            bd.mean = bd.mean + (-lr * Fg) * adaptive_dt + mean_noise
            H = -(1.0 / cfg.noise_scale ** 2 + 1.0) * torch.eye(L, dtype=torch.float64)
            bd.covariance = bd._matrix_exponential_update(H, dt, D)
            bd.covariance = bd._ensure_numerical_stability(bd.covariance)
            bd.precision = bd._safe_inverse(bd.covariance)
            ent = float(bd.entropy())
            # ---- oracle
            mean_o, cov_o, prec_o, cond = orc.fokker_planck_full_step(mean_o, cov_o, obs, sc, xi, dt=dt, D=D, lr=lr, noise_scale=cfg.noise_scale, min_var=cfg.min_variance)
            scale = np.abs(bd.covariance.numpy()).max()
            np.testing.assert_allclose(mean_o, bd.mean.numpy(), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(cov_o, bd.covariance.numpy(), rtol=1e-8, atol=1e-10 * scale)
            np.testing.assert_allclose(prec_o, bd.precision.numpy(), rtol=1e-6, atol=1e-9 * np.abs(prec_o).max())
            np.testing.assert_allclose(orc.gaussian_entropy_full(cov_o), ent, rtol=1e-9)
            np.testing.assert_allclose(cond, bd.history["condition_numbers"][-1], rtol=1e-6)
            for k, v in (("obs", obs), ("score", sc), ("xi", xi), ("mean", bd.mean.numpy().copy()), ("cov", bd.covariance.numpy().copy()),
                         ("prec", bd.precision.numpy().copy()), ("entropy", np.float64(ent)), ("cond", np.float64(cond))):
                out[f"{name}_{k}{t}"] = v
        print(name, "ok: L", L, "steps", steps, "cond", [float(x) for x in bd.history["condition_numbers"]])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fp_full.npz"), **out)
    print("wrote tests/golden/fp_full.npz", sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
