"""Metric tolerances shared by the whole-step parity tests."""
LOSS_KEYS = ("reconstruction_loss", "kl_loss", "score_matching_loss", "elbo", "reward_loss", "grad_penalty", "policy_loss", "value_loss",
             "dynamics_loss", "mean_time", "loss_weight_mean", "pragmatic_mean", "consistency_mean", "contrastive_loss")


def assert_metric(k, got, want, tag=""):
    """Losses at rtol 1e-4 (+1e-6 for values that pass through zero); the estimator's statistics are differences of O(1) batch means of
    a wide net, held to 1e-4 relative + 2e-5 absolute (the bound tests/test_hip_headline.py uses at the headline size)."""
    tol = 1e-4 * abs(want) + (1e-6 if k in LOSS_KEYS else 2e-5)
    assert abs(got - want) <= tol, f"{tag} {k}: got {got!r}, want {want!r}, |diff| {abs(got - want):.3e} > {tol:.3e}"
