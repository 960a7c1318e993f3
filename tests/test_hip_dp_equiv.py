"""GPU: N ranks x B rows == 1 rank x N*B rows (SURVEY section 4 item 5, 8(e) "disjoint index slices of one global draw").

Two gloo ranks sharing cuda:0 each train on their slice of ONE global index draw and ONE global noise tape; a single process trains on
the whole 2B-row batch with the same tape.  After two train_steps the four optimiser arenas, the last gradients, the reward
normaliser and the time-importance weights agree.  What makes this exact rather than statistical: gradient means over ranks (one
all-reduce per arena), the reward normaliser and the KL annealing weight exp(-5 mean t) on the all-gathered batch, the sequential
time-importance EMA replayed in rank-major order, and a 5-row lambda-return halo from the next rank (aid_value_step_ex).  The MINE
statistics are per rank (gradient-free): the EFE value / epistemic metrics are not compared."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

O, A, D, N, H, B, W, CAP = 17, 6, 64, 4, 2, 8, 2, 64
KEY, WKEY = 777, 41
ARENAS = ("score", "policy", "value", "dynamics")
LINEAR_METRICS = ("reconstruction_loss", "score_matching_loss", "reward_loss", "grad_penalty", "kl_loss", "value_loss", "dynamics_loss")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Box:
    def __init__(self, shape):
        self.shape = shape


def _store():
    from oracle.noise import NoiseStream
    bs = NoiseStream(5)
    return {"observations": bs.randn(CAP, O), "actions": np.tanh(bs.randn(CAP, A)).astype(np.float32), "rewards": bs.randn(CAP),
            "next_observations": bs.randn(CAP, O), "dones": bs.rand(CAP) < 0.15}


def _make_agent(batch, prefetch=False):
    import aid_amd
    from oracle import aid_oracle as orc
    from oracle.weights import draw_params, param_specs
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=batch, efe_horizon=H, latent_dim=O, hidden_dim=D)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)

    class Env:
        observation_space, action_space = Box((O,)), Box((A,))
    agent = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=CAP))
    ai = agent.active_inference
    p = draw_params(param_specs(O, A, O, D, N), WKEY, orc.schedule_tables(N))
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    st = _store()
    agent.replay_buffer.add_batch(st["observations"], st["actions"], st["rewards"], st["next_observations"], st["dones"])
    # prefetch off: one noise call per phase.  Prefetch on: the reference's draw order with the look-ahead ELBO (shadow reward normaliser
    # committed after reward_normalize_global, the score all-reduce issued one step late) - the same order on both sides too
    agent.prefetch_next_batch = prefetch
    return agent


def _sharded_noise(rank, world):
    """The global tape (ParityNoise at batch world*B) cut into this rank's rows."""
    from aid_amd.noise import NoiseSource
    from parity_noise import ParityNoise

    class Sharded(NoiseSource):
        def __init__(self):
            super().__init__(seed=0)
            self.g = ParityNoise(KEY)

        def _sl(self, b):
            return slice(rank * b, (rank + 1) * b)

        def replay_indices(self, size, batch):
            return self.g.replay_indices(size, batch * world)[self._sl(batch)]

        def belief(self, b, L, Dh, n, train):
            g, sl = self.g.belief(b * world, L, Dh, n, train), self._sl(b)
            return {"z_init": g["z_init"][sl].contiguous(), "step_noise": g["step_noise"][:, sl].contiguous(),
                    "drop_masks": g["drop_masks"][:, sl].contiguous() if g["drop_masks"] is not None else None,
                    "dec_masks": [m[sl].contiguous() for m in g["dec_masks"]] if g["dec_masks"] is not None else None}

        def elbo(self, b, L, Dh, first, tiw):
            g, sl = self.g.elbo(b * world, L, Dh, first, tiw), self._sl(b)
            return {k: v[sl].contiguous() for k, v in g.items()}

        def efe(self, b, L, Aa, Hh, K, S):
            g, sl, bg = self.g.efe(b * world, L, Aa, Hh, K, S), self._sl(b), b * world
            out = {"policy_eps": g["policy_eps"].view(Hh, K, bg, Aa)[:, :, sl].reshape(Hh, K * b, Aa).contiguous(),
                   "reparam_eps": g["reparam_eps"].view(Hh, K, bg, L)[:, :, sl].reshape(Hh, K * b, L).contiguous(),
                   "mine_z_eps": g["mine_z_eps"].view(Hh, K, S, bg, L)[:, :, :, sl].reshape(Hh, K * S * b, L).contiguous(),
                   "mine_delta": g["mine_delta"].view(4, Hh, K, S, bg, L)[:, :, :, :, sl].reshape(4, Hh, K * S * b, L).contiguous(),
                   "proj_mask": g["proj_mask"].view(Hh, K, S, bg, 512)[:, :, :, sl].reshape(Hh, K * S * b, 512).contiguous(),
                   "mine_mask": g["mine_mask"].view(4, Hh, K, S, bg, 512)[:, :, :, :, sl].reshape(4, Hh, K * S * b, 512).contiguous()}
            rng = np.random.default_rng(rank)        # row permutations of the gradient-free estimator: any valid ones
            out["perms"] = torch.from_numpy(np.stack([rng.permutation(b) for _ in range(Hh * K * S)]).reshape(Hh, K, S, b).astype(np.int32)).cuda()
            return out
    return Sharded()


def _snapshot(agent, metrics):
    ai = agent.active_inference
    torch.cuda.synchronize()
    return {"flat": {n: ai.arenas[n].flat.detach().cpu().numpy().copy() for n in ARENAS},
            "grad": {n: ai.arenas[n].grad.detach().cpu().numpy().copy() for n in ARENAS},
            "rms": agent.reward_normalizer.state.cpu().numpy().copy(), "tiw": ai.time_importance_weights.cpu().numpy().copy(),
            "metrics": metrics}


def _worker(rank, world, port, q, prefetch, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        agent = _make_agent(B, prefetch)
        agent.enable_data_parallel()
        sn = _sharded_noise(rank, world)
        agent.active_inference.noise = sn; agent.replay_buffer.index_source = sn
        ms = [agent.train_step() for _ in range(steps)]
        assert (agent._lookahead is not None) == prefetch
        q.put((rank, "ok", _snapshot(agent, ms)))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("prefetch,steps", [(False, 2), (True, 3)])
def test_two_ranks_equal_one_rank_at_twice_the_batch(prefetch, steps):
    from parity_noise import ParityNoise
    # ---- the single-process run on the global batch
    agent = _make_agent(W * B, prefetch)
    p0 = {n: agent.active_inference.arenas[n].flat.detach().cpu().numpy().copy() for n in ARENAS}
    pn = ParityNoise(KEY); agent.active_inference.noise = pn; agent.replay_buffer.index_source = pn
    ref = _snapshot(agent, [agent.train_step() for _ in range(steps)])
    del agent, pn; torch.cuda.empty_cache()
    # ---- two ranks, B rows each
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, W, port, q, prefetch, steps)) for r in range(W)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda x: x[0])
    for p in procs: p.join(60)
    for r, msg, _ in res:
        assert msg == "ok", f"rank {r}:\n{msg}"
    snaps = [s for _, _, s in res]
    for n in ARENAS:                                         # replicas identical
        np.testing.assert_array_equal(snaps[0]["flat"][n], snaps[1]["flat"][n])
    got = snaps[0]
    lr = 5e-5
    for n in ARENAS:
        # the (averaged, unclipped) gradients of the last step: summation order differs (two halves vs one batch), nothing else
        g_ref, g_got = ref["grad"][n].astype(np.float64), got["grad"][n].astype(np.float64)
        if prefetch and n == "score":
            # the score arena now holds the LOOK-AHEAD ELBO's gradients (the next step's), whose all-reduce is issued at the start of the next
            # step: each rank still has its local half - their mean is what the collective will produce
            g_got = np.mean([s_["grad"][n].astype(np.float64) for s_ in snaps], axis=0)
        rel = np.linalg.norm(g_got - g_ref) / max(np.linalg.norm(g_ref), 1e-30)
        assert rel < 1e-4, (n, "gradient", rel)
        # parameters after the steps: rtol 1e-5 except the few elements whose |g| ~ Adam's eps (a different step of size <= lr)
        a, b = got["flat"][n].astype(np.float64), ref["flat"][n].astype(np.float64)
        err = np.abs(a - b)
        viol = err > 1e-5 * np.abs(b) + 1e-7
        assert err.max() <= 2 * steps * lr + 1e-6 and viol.sum() <= max(2, 1e-4 * a.size), (n, err.max(), int(viol.sum()), a.size)
        upd = np.linalg.norm((a - p0[n]) - (b - p0[n])) / max(np.linalg.norm(b - p0[n]), 1e-30)
        assert upd < 1e-2, (n, "update", upd)
    np.testing.assert_allclose(got["rms"], ref["rms"], rtol=1e-12)
    np.testing.assert_allclose(got["tiw"], ref["tiw"], rtol=1e-5, atol=1e-7)
    for step in range(steps):      # batch means: the mean over ranks of the per-rank value is the global-batch value
        for k in LINEAR_METRICS:
            dp_mean = np.mean([s["metrics"][step][k] for s in snaps])
            np.testing.assert_allclose(dp_mean, ref["metrics"][step][k], rtol=2e-4, atol=1e-6, err_msg=f"step {step} {k}")
        assert all(abs(s["metrics"][step]["mean_time"] - ref["metrics"][step]["mean_time"]) < 1e-6 for s in snaps)   # global mean on every rank


def test_lambda_return_halo_confines_the_seam():
    """compute_lambda_returns reads rows idx..idx+5 of the minibatch (active_inference.py:638-707).  A shard without its successor's rows
    differs from the global batch in its last 5 rows at most, and nowhere once the 5-row halo is appended."""
    from aid_amd import _native as nat
    rng = np.random.default_rng(0)
    n, b = 64, 32
    dev = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda()
    r, nv, d = dev(rng.standard_normal(n)), dev(rng.standard_normal(n)), dev(rng.random(n) < 0.1)
    full = nat.lambda_returns(r, nv, d, 0.99, 0.95, 5)
    shard = nat.lambda_returns(r[:b].contiguous(), nv[:b].contiguous(), d[:b].contiguous(), 0.99, 0.95, 5)
    diff = (shard != full[:b]).nonzero().flatten().tolist()
    assert diff and min(diff) >= b - 5, diff
    halo = nat.lambda_returns(r[:b + 5].contiguous(), nv[:b + 5].contiguous(), d[:b + 5].contiguous(), 0.99, 0.95, 5)
    assert torch.equal(halo[:b], full[:b])
    last = nat.lambda_returns(r[b:].contiguous(), nv[b:].contiguous(), d[b:].contiguous(), 0.99, 0.95, 5)
    assert torch.equal(last, full[b:])                          # the last rank needs no halo
