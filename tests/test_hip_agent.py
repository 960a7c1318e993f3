"""GPU: the drop-in surface (DiffusionStateAgent.train_step / act, state_dict layout) against the reference's own
numbers (tests/golden, two full train_steps with K=10, S=10) and against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import aid_oracle as orc
from oracle.noise import NoiseStream
from oracle.weights import draw_params, param_specs
from parity_noise import ParityNoise
from tolerances import assert_metric


class Box:
    def __init__(self, shape):
        self.shape = shape


class FakeEnv:
    def __init__(self, O, A):
        self.observation_space, self.action_space = Box((O,)), Box((A,))


def _agent(O, A, D, N, H, B, cap=64, L=None):
    import aid_amd
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=H, latent_dim=L or O, hidden_dim=D)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)
    return aid_amd.DiffusionStateAgent(FakeEnv(O, A), cfg, aid_amd.TrainingConfig(buffer_size=cap))


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_state_dict_layout_and_golden_train_steps(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"state_{tag}.npz"))
    O, A, L, D, N, B, H = (int(g[k]) for k in ("O", "A", "L", "D", "N", "B", "H"))
    agent = _agent(O, A, D, N, H, B)
    ai = agent.active_inference
    specs = param_specs(O, A, L, D, N)
    sd = ai.state_dict()
    names = [k for k in sd if not k.startswith("epistemic_estimator.decoder.")]
    assert names == [n for n, _ in specs]                                   # reference key order
    assert all(tuple(sd[n].shape) == s for n, s in specs)
    assert sum(1 for k in sd if k.startswith("epistemic_estimator.decoder.")) == 14   # aliases of observation_decoder.*
    p = draw_params(specs, int(g["weights_key"]), orc.schedule_tables(N))
    full = dict(p); full.update({k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd if k.startswith("epistemic_estimator.decoder.")})
    ai.load_state_dict(full, strict=True)
    bs = NoiseStream(int(g["store_key"])); cap = int(g["store_cap"])
    store = {"observations": bs.randn(cap, O), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32), "rewards": bs.randn(cap),
             "next_observations": bs.randn(cap, O), "dones": bs.rand(cap) < 0.1}
    for i in range(cap):
        agent.replay_buffer.add(store["observations"][i], store["actions"][i], float(store["rewards"][i]), store["next_observations"][i], bool(store["dones"][i]))
    pn = ParityNoise(0)
    ai.noise = pn; agent.replay_buffer.index_source = pn
    agent.prefetch_next_batch = False      # the goldens re-key the noise stream per train_step; the prefetch draws across step boundaries
    for step in (1, 2):
        pn.rekey(int(g[f"ts{step}_key"]))
        m = agent.train_step()
        for k in g.files:
            if k.startswith(f"ts{step}_metric/"):
                name = k.split("/", 1)[1]
                if name in m:
                    assert_metric(name, float(m[name]), float(g[k]), f"step {step}")
        sd = ai.state_dict()
        for k in g.files:
            if k.startswith(f"ts{step}_param/"):
                np.testing.assert_allclose(sd[k.split("/", 1)[1]].cpu().numpy(), g[k], rtol=1e-3, atol=2e-6, err_msg=k)
        # every parameter, via checksums of the reference's post-step state_dict
        cs = np.array([float(sd[n].double().sum()) for n, _ in specs])
        tol = 2e-5 * g[f"ts{step}_abssum"] + 2e-5
        # the estimator's running mean is an EMA of mean(exp(T)) with T ~ 20-40 at these random weights (values ~1e9): an absolute fp32
        # rounding difference of 2e-5 in T is a RELATIVE difference of 2e-5 in exp(T), so this one entry is held to 1e-4 relative
        tol[[n for n, _ in specs].index("epistemic_estimator.running_mean")] *= 5.0
        bad = np.abs(cs - g[f"ts{step}_checksum"]) > tol
        assert not bad.any(), [specs[i][0] for i in np.nonzero(bad)[0]]
        np.testing.assert_allclose(agent.score_ema.shadow["output_proj.2.weight"].cpu().numpy(),
                                   g[f"ts{step}_ema/latent_score_network.output_proj.2.weight"], rtol=1e-4, atol=1e-7)
    # decoder / reward predictor / estimator never move
    for n in ("observation_decoder.3.weight", "reward_predictor.5.weight"):
        np.testing.assert_array_equal(sd[n].cpu().numpy(), p[n].numpy())


def test_train_step_full_width_vs_oracle():
    """D=512 (the BASELINE width), small batch/horizon so the CPU oracle finishes in seconds: one whole train_step."""
    O, A, D, N, H, B = 17, 6, 512, 3, 2, 16
    agent = _agent(O, A, D, N, H, B)
    ai = agent.active_inference
    specs = param_specs(O, A, O, D, N)
    p = draw_params(specs, 31, orc.schedule_tables(N))
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    bs = NoiseStream(8); cap = 64
    store = {"observations": bs.randn(cap, O), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32), "rewards": bs.randn(cap),
             "next_observations": bs.randn(cap, O), "dones": bs.rand(cap) < 0.1}
    agent.replay_buffer.add_batch(store["observations"], store["actions"], store["rewards"], store["next_observations"], store["dones"])
    pn = ParityNoise(555); ai.noise = pn; agent.replay_buffer.index_source = pn
    m = agent.train_step()
    st = orc.OracleAgentState(p, N)
    ns = NoiseStream(555)
    m_ref = orc.train_step(st, orc.replay_sample(store, cap, B, ns), ns, horizon=H)
    for k, v in m_ref.items():
        if k in m and isinstance(v, float):
            assert_metric(k, float(m[k]), v)
    sd = ai.state_dict()
    for n, _ in specs:
        if n == "epistemic_estimator.running_mean":
            np.testing.assert_allclose(float(sd[n]), st.running_mean, rtol=1e-3); continue
        ref = st.p[n].detach().numpy()
        np.testing.assert_allclose(sd[n].cpu().numpy(), ref, rtol=1e-3, atol=2e-6, err_msg=n)


def test_act_and_philox_train_step_run():
    """Production mode (device Philox): act() returns a finite action of the right shape; train_step returns every
    reference metric key and changes exactly the four trained groups."""
    O, A = 17, 6
    agent = _agent(O, A, 64, 4, 2, 8, cap=32)
    assert agent.train_step() == {}                      # buffer smaller than the batch (state_agent.py:112-113)
    rng = np.random.default_rng(0)
    for _ in range(32):
        agent.replay_buffer.add(rng.standard_normal(O), rng.uniform(-1, 1, A), float(rng.standard_normal()), rng.standard_normal(O), False)
    before = {k: v.clone() for k, v in agent.active_inference.state_dict().items()}
    m = agent.train_step(); m = agent.train_step(); m = agent.train_step()
    want = {"reconstruction_loss", "kl_loss", "score_matching_loss", "elbo", "reward_loss", "grad_penalty", "mean_time", "loss_weight_mean",
            "policy_loss", "epistemic_mean", "pragmatic_mean", "consistency_mean", "num_trajectories", "horizon", "epistemic/mi_estimate",
            "epistemic/joint_term", "epistemic/marginal_term", "epistemic/running_mean", "value_loss", "dynamics_loss"}
    assert want <= set(m) and all(np.isfinite(v) for v in m.values())
    after = agent.active_inference.state_dict()
    moved = {k.split(".")[0] for k in after if after[k].dtype.is_floating_point and not torch.equal(after[k], before[k])}
    assert {"latent_score_network", "policy_network", "value_network", "latent_dynamics"} <= moved
    assert "observation_decoder" not in moved and "reward_predictor" not in moved
    action, info = agent.act(rng.standard_normal(O).astype(np.float32))
    assert action.shape == (A,) and np.all(np.isfinite(action)) and np.all(np.abs(action) <= 1)
    assert np.isfinite(info["expected_free_energy"]) and info["latent"].shape == (1, O)
    import aid_amd.agents as ag
    assert ag.StateBasedAgent is ag.DiffusionStateAgent
    # opt-in estimator training every 5th step (state_agent.py:216-222; raises in the reference)
    w0 = agent.active_inference.epistemic_estimator.mine_network[0].weight.detach().clone()
    agent.train_epistemic_every = 5; agent.total_steps = 5
    m = agent.train_step()
    assert {"epistemic_mi", "epistemic/mi_estimate", "epistemic/running_mean"} <= set(m) and np.isfinite(m["epistemic_mi"])
    assert not torch.equal(w0, agent.active_inference.epistemic_estimator.mine_network[0].weight)
    m = agent.train_step(); assert "epistemic_mi" not in m


def test_cross_step_prefetch_equals_reference_order():
    """Three back-to-back train_steps with the next-minibatch prefetch ON (step i runs its third belief update together with step
    i+1's two, on the side stream) against the oracle stepping in the reference's order on ONE continuous noise stream; then the
    guards: an insertion into the replay buffer or a load_state_dict makes the next step discard the prefetched latents."""
    O, A, D, N, H, B = 17, 6, 64, 4, 2, 8
    agent = _agent(O, A, D, N, H, B)
    ai = agent.active_inference
    specs = param_specs(O, A, O, D, N)
    p = draw_params(specs, 77, orc.schedule_tables(N))
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    bs = NoiseStream(9); cap = 64
    store = {"observations": bs.randn(cap, O), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32), "rewards": bs.randn(cap),
             "next_observations": bs.randn(cap, O), "dones": bs.rand(cap) < 0.1}
    agent.replay_buffer.add_batch(store["observations"], store["actions"], store["rewards"], store["next_observations"], store["dones"])
    pn = ParityNoise(4242); ai.noise = pn; agent.replay_buffer.index_source = pn
    assert agent.prefetch_next_batch
    st = orc.OracleAgentState(p, N); ns = NoiseStream(4242)
    for step in range(3):
        m = agent.train_step()
        assert (agent._prefetch is not None) and (step == 0 or agent._prefetch_backoff == 0)
        # the next step's ELBO has been enqueued on the prefetched minibatch (look-ahead) - on shadow state: what the agent shows between
        # steps is the reference's state after `step + 1` updates, not one batch ahead
        assert agent.lookahead_elbo and agent._lookahead is not None and agent._lookahead["pre"] is agent._prefetch
        assert abs(agent.reward_normalizer.count - (1e-4 + (step + 1) * B)) < 1e-6
        assert abs(float(agent._lookahead["rn_state"][2]) - (1e-4 + (step + 2) * B)) < 1e-6
        m_ref = orc.train_step(st, orc.replay_sample(store, cap, B, ns), ns, horizon=H)
        for k, v in m_ref.items():
            if k in m and isinstance(v, float):
                assert_metric(k, float(m[k]), v, f"step {step}")
    sd = ai.state_dict()
    for n, _ in specs:
        if n == "epistemic_estimator.running_mean":
            continue
        np.testing.assert_allclose(sd[n].cpu().numpy(), st.p[n].detach().numpy(), rtol=2e-3, atol=3e-6, err_msg=n)
    np.testing.assert_allclose(agent.reward_normalizer.state.cpu().numpy(), [st.rms.mean, st.rms.var, st.rms.count], rtol=1e-6)   # fp32 rewards in, fp64 statistics
    tiw_before = ai.time_importance_weights.clone()
    # guards
    agent.replay_buffer.add(store["observations"][0], store["actions"][0], 0.0, store["next_observations"][0], False)
    assert agent._take_prefetch(B) is None and agent._prefetch_wait == 1
    assert agent._lookahead is not None
    count_before = agent.reward_normalizer.count
    agent.train_step()                                   # backoff: this step does not prefetch; the look-ahead ELBO is dropped, its shadow state never committed
    assert agent._prefetch is None and agent._lookahead is None
    assert abs(agent.reward_normalizer.count - (count_before + B)) < 1e-6 and not torch.equal(ai.time_importance_weights, tiw_before)
    agent.train_step(); assert agent._prefetch is not None
    ai.load_state_dict(ai.state_dict())
    assert agent._take_prefetch(B) is None


def test_overlap_streams_toggled_between_steps_without_host_sync():
    """The look-ahead ELBO of step i+1 is still running on the side stream when step i returns.  Switching `overlap_streams` off before
    step i+1 (no device sync in between) must not let the main stream commit / step on gradients that are not written yet: four steps,
    toggling after every one, equal the oracle stepping in the reference's order."""
    O, A, D, N, H, B = 17, 6, 64, 4, 2, 8
    agent = _agent(O, A, D, N, H, B)
    ai = agent.active_inference
    specs = param_specs(O, A, O, D, N)
    p = draw_params(specs, 78, orc.schedule_tables(N))
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    bs = NoiseStream(10); cap = 64
    store = {"observations": bs.randn(cap, O), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32), "rewards": bs.randn(cap),
             "next_observations": bs.randn(cap, O), "dones": bs.rand(cap) < 0.1}
    agent.replay_buffer.add_batch(store["observations"], store["actions"], store["rewards"], store["next_observations"], store["dones"])
    pn = ParityNoise(4343); ai.noise = pn; agent.replay_buffer.index_source = pn
    st = orc.OracleAgentState(p, N); ns = NoiseStream(4343)
    took_la = 0
    for step in range(4):
        agent.overlap_streams = step % 2 == 0            # on, off, on, off - the off steps inherit a look-ahead ELBO from the on steps
        took_la += int(agent._lookahead is not None and not agent.overlap_streams)
        m = agent.train_step()
        m_ref = orc.train_step(st, orc.replay_sample(store, cap, B, ns), ns, horizon=H)
        for k, v in m_ref.items():
            if k in m and isinstance(v, float):
                assert_metric(k, float(m[k]), v, f"step {step}")
    assert took_la == 2
    sd = ai.state_dict()
    for n, _ in specs:
        if n != "epistemic_estimator.running_mean":
            np.testing.assert_allclose(sd[n].cpu().numpy(), st.p[n].detach().numpy(), rtol=2e-3, atol=3e-6, err_msg=n)


def test_graphed_small_batch_denoise_matches_eager():
    """act()-sized reverse loops are replayed from a captured hipGraph: same latents as the eager loop fed the same Philox draws, still
    valid after an optimiser step (weights are read in place), and per-call randomness differs."""
    from aid_amd import _native as nat
    O, A = 17, 6
    agent = _agent(O, A, 64, 6, 2, 8, cap=32)
    ai = agent.active_inference
    rng = np.random.default_rng(1)
    for _ in range(32):
        agent.replay_buffer.add(rng.standard_normal(O), rng.uniform(-1, 1, A), float(rng.standard_normal()), rng.standard_normal(O), False)
    obs = torch.randn(4, O, device="cuda")
    for train in (False, True):
        ai.train(train)
        for rep in range(2):
            off0 = ai.noise._offset
            z_graph = ai.denoise_only(obs)
            # eager loop with the very draws the graphed call made
            N, L, D = 6, O, 64
            z0 = torch.empty(4, L, device="cuda"); eps = torch.empty(N, 4, L, device="cuda")
            o = off0; nat.fill_normal(z0, ai.noise.seed, o); o += z0.numel(); nat.fill_normal(eps, ai.noise.seed, o); o += eps.numel()
            mask = None
            if train:
                mask = torch.empty(N, 4, D, device="cuda"); nat.fill_dropout_mask(mask, 0.1, ai.noise.seed, o)
            z_eager = nat.denoise_loop(ai.latent_score_network.table(), ai.latent_diffusion.coefs(), obs, N, z0, eps, mask, train=train)
            assert torch.equal(z_graph, z_eager)
            if rep == 0:
                first = z_graph.clone()
                ai.train(True); agent.train_step(); ai.train(train)          # parameters move; the captured graph must see them
            else:
                assert not torch.equal(first, z_graph)
    assert len(ai._denoise_graphs) == 2


@pytest.mark.parametrize("O,A,B,D,N,H", [(11, 3, 100, 128, 5, 3), (23, 7, 36, 64, 4, 1)])
def test_ragged_sizes_train_and_act(O, A, B, D, N, H):
    """Batch sizes / dims that are not multiples of any tile (100 rows -> 300-row merged loop, latent 11 or 23): three train_steps
    (device Philox, prefetch + side stream on) stay finite and act() works."""
    agent = _agent(O, A, D, N, H, B, cap=4 * B)
    rng = np.random.default_rng(O)
    agent.replay_buffer.add_batch(rng.standard_normal((4 * B, O)), rng.uniform(-1, 1, (4 * B, A)), rng.standard_normal(4 * B),
                                  rng.standard_normal((4 * B, O)), rng.random(4 * B) < 0.05)
    for _ in range(3):
        m = agent.train_step()
        assert all(np.isfinite(v) for v in m.values()), m
    a, info = agent.act(rng.standard_normal(O).astype(np.float32))
    assert a.shape == (A,) and np.all(np.isfinite(a))


def test_state_mode_latent_dim_differs_from_obs_dim():
    """State mode with latent_dim != observation_dim (the config default is 128 on 17-d observations): the reference cannot run it
    (its obs encoder is sized latent_dim but fed raw states), so there is no parity to pin; the build sizes the obs encoder
    with the true observation dim.  Three train_steps stay finite, the decoder maps latents back to O dims, act() works."""
    O, A, L, B = 17, 6, 32, 32
    agent = _agent(O, A, 64, 4, 2, B, cap=4 * B, L=L)
    rng = np.random.default_rng(5)
    agent.replay_buffer.add_batch(rng.standard_normal((4 * B, O)), rng.uniform(-1, 1, (4 * B, A)), rng.standard_normal(4 * B),
                                  rng.standard_normal((4 * B, O)), rng.random(4 * B) < 0.05)
    for _ in range(3):
        m = agent.train_step()
        assert all(np.isfinite(v) for v in m.values()), m
    assert agent._last["latents"].shape == (B, L)
    assert agent.active_inference.decode_observation(agent._last["latents"]).shape == (B, O)
    a, info = agent.act(rng.standard_normal(O).astype(np.float32))
    assert a.shape == (A,) and np.all(np.isfinite(a))


def _layout_names(opt, agent):
    """Our optimiser's checkpoint layout as the reference spells its parameters: 'active_inference.<name>' / 'encoder.<name>' or None."""
    out = []
    for e in opt.layout:
        if e is None:
            out.append(None)
        else:
            n = e[1]
            out.append(n if n.startswith("encoder.") else "active_inference." + n)
    return out


@pytest.mark.parametrize("mode", ["state", "pixel"])
def test_optimizer_checkpoint_layout_matches_reference(golden_dir, mode):
    """torch.optim.AdamW indexes its state by position in the parameter list it was given, so `*_optimizer.state_dict()` exchanges
    with the reference's checkpoints (utils/training.py:72-172) only if every optimiser lists the same parameters in the same order:
    tests/golden/opt_layout.json holds the reference's four lists for both agents (names + shapes).  Positions the reference lists but
    never updates (observation decoder, reward predictor, the encoder's dead channel_reduce conv) carry no state on either side."""
    import json
    import aid_amd
    g = json.load(open(os.path.join(golden_dir, "opt_layout.json")))[mode]
    d = g["dims"]
    if mode == "state":
        agent = _agent(d["O"], d["A"], d["D"], d["N"], 2, 8)
    else:
        cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=3, efe_horizon=1, latent_dim=d["L"], hidden_dim=d["D"], pixel_observation=True)
        cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=d["N"])

        class Env:
            observation_space, action_space = Box((3, 3, 84, 84)), Box((d["A"],))
        agent = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=4), aid_amd.PixelObservationConfig())
    for key in ("score_optimizer", "policy_optimizer", "value_optimizer", "dynamics_optimizer"):
        opt, ref = getattr(agent, key), g[key]
        ours = _layout_names(opt, agent)
        assert len(ours) == len(ref), (key, len(ours), len(ref))
        for i, (o, (rn, rshape)) in enumerate(zip(ours, ref)):
            if o is None:        # listed by the reference, never stepped: no state on either side
                assert rn.startswith(("active_inference.observation_decoder.", "active_inference.reward_predictor.", "encoder.attention.channel_reduce")), (key, i, rn)
                continue
            assert o == rn, (key, i, o, rn)
            arena, name = opt.layout[i]
            assert list(arena.offsets[name][1]) == rshape, (key, name)
        sd = opt.state_dict()
        assert sd["param_groups"][0]["params"] == list(range(len(ref))) and sd["state"] == {}        # nothing stepped yet
    assert agent.value_optimizer.state_dict()["param_groups"][0]["weight_decay"] == 1e-2 and agent.score_optimizer.state_dict()["param_groups"][0]["weight_decay"] == 1e-5


def test_concurrent_belief_loops_equal_the_merged_loop():
    """Three independent belief updates as three concurrent reverse loops (DiffusionActiveInference._denoise_chains: one stream each, the
    process-wide chain streams of aid_amd/streams.py) against the one merged loop over the concatenated rows, on the same explicit noise:
    the rows of a loop never interact, so the latents agree to fp32 rounding (a layer's tile height follows the row count, its K order
    does not).  Also: the chains are skipped (merged loop) when an entry is too small or too large for them."""
    from aid_amd import streams
    O, A, D, N, H, B = 17, 6, 64, 4, 2, 160
    agent = _agent(O, A, D, N, H, B, cap=256)
    ai = agent.active_inference
    rng = np.random.default_rng(5)
    obs = [torch.from_numpy(rng.standard_normal((B, O)).astype(np.float32)).cuda() for _ in range(3)]
    pn = ParityNoise(31)
    noises = [pn.belief(B, ai.latent_dim, D, N, True) for _ in range(3)]
    ai.concurrent_belief_loops = False
    merged = ai.denoise_many(obs, noises)
    ai.concurrent_belief_loops = True
    assert ai._chain_streams_for(3) is not None and len(streams.chain_streams()) == 2
    chains = ai.denoise_many(obs, noises)
    torch.cuda.synchronize()
    assert chains.shape == merged.shape == (3 * B, ai.latent_dim)
    np.testing.assert_allclose(chains.cpu().numpy(), merged.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # from a side stream: the calling stream is chain 0, the process's two chain streams carry the others
    side = streams.side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        again = ai.denoise_many(obs, noises)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(again, chains)
    # entries outside (128, CHAIN_MAX_ROWS] rows, or more entries than chains: one merged loop
    small = [o[:64] for o in obs]
    assert not all(128 < o.shape[0] <= ai.CHAIN_MAX_ROWS for o in small)
    assert ai._chain_streams_for(4) is None


@pytest.mark.parametrize("B", [8, 160])
def test_dead_next_belief_elision_changes_nothing(B):
    """state_agent.py:138-139 computes a next_obs belief that :195-196 overwrites before any read.  With `elide_dead_next_belief` the agent does
    not run that reverse loop (it only consumes its draws): three back-to-back train_steps on one parity tape with the switch on and off -
    unprefetched first step, prefetched steps with the look-ahead ELBO, merged loops (B = 8) and concurrent chains (B = 160) - give the same
    metrics, the same post-step state_dict, the same normaliser / time-importance state, and the tape ends at the same position."""
    O, A, D, N, H = 17, 6, 64, 4, 2
    cap = 256
    specs = param_specs(O, A, O, D, N)
    p = draw_params(specs, 78, orc.schedule_tables(N))
    bs = NoiseStream(10)
    store = {"observations": bs.randn(cap, O), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32), "rewards": bs.randn(cap),
             "next_observations": bs.randn(cap, O), "dones": bs.rand(cap) < 0.1}
    runs = {}
    for elide in (False, True):
        agent = _agent(O, A, D, N, H, B, cap=cap)
        agent.elide_dead_next_belief = elide
        ai = agent.active_inference
        sd0 = ai.state_dict()
        ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
        agent.replay_buffer.add_batch(store["observations"], store["actions"], store["rewards"], store["next_observations"], store["dones"])
        pn = ParityNoise(555); ai.noise = pn; agent.replay_buffer.index_source = pn
        ms = [agent.train_step() for _ in range(3)]
        assert agent._prefetch is not None and agent._prefetch["latents"].shape[0] == (B if elide else 2 * B)
        torch.cuda.synchronize()
        runs[elide] = {"metrics": ms, "sd": {k: v.detach().cpu().clone() for k, v in ai.state_dict().items()},
                       "rn": agent.reward_normalizer.state.cpu().clone(), "tiw": ai.time_importance_weights.cpu().clone(),
                       "tape": float(pn.ns.randn(1, 1)[0, 0]), "last": {k: v.cpu().clone() for k, v in agent._last.items()}}
    a, b = runs[False], runs[True]
    assert a["tape"] == b["tape"]                       # both consumed exactly the reference's draws
    for m0, m1 in zip(a["metrics"], b["metrics"]):
        assert set(m0) == set(m1)
        for k in m0:
            # The unprefetched first step runs the obs loop as a direct launch sequence on the caller's stream with the switch on and as a
            # hipGraph captured on a chain stream (low-LDS GEMM variants: another summation order) with it off: latents agree to fp32 rounding.
            # The MINE statistics are exp(T) with T ~ 70 under these random weights: 3 ulp of T are 2e-5 of exp(T).
            np.testing.assert_allclose(m1[k], m0[k], rtol=2e-4 if k.startswith("epistemic") else 1e-5, atol=1e-7, err_msg=k)
    for k in a["sd"]:
        np.testing.assert_allclose(b["sd"][k].numpy(), a["sd"][k].numpy(), rtol=2e-4 if k.endswith("running_mean") else 1e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(b["rn"].numpy(), a["rn"].numpy(), rtol=1e-12)
    np.testing.assert_allclose(b["tiw"].numpy(), a["tiw"].numpy(), rtol=1e-5, atol=1e-7)
    for k in a["last"]:
        np.testing.assert_allclose(b["last"][k].numpy(), a["last"][k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_denoise_start_join_equals_the_paired_loop():
    """DiffusionActiveInference.denoise_start / denoise_join (what the pixel agent uses to run the belief over obs underneath encode(next_obs) and the one over
    next_obs underneath the ELBO): two separately started chains on the process's chain streams give the latents of the paired loop on the same explicit
    noise; outside (128, CHAIN_MAX_ROWS] rows, or with concurrent loops switched off, the caller is told to take the merged loop (None)."""
    O, A, D, N, H, B = 17, 6, 64, 4, 2, 160
    agent = _agent(O, A, D, N, H, B, cap=256)
    ai = agent.active_inference
    rng = np.random.default_rng(6)
    obs = [torch.from_numpy(rng.standard_normal((B, O)).astype(np.float32)).cuda() for _ in range(2)]
    pn = ParityNoise(32)
    noises = [pn.belief(B, ai.latent_dim, D, N, True) for _ in range(2)]
    ai.concurrent_belief_loops = True
    pair = ai.denoise_pair(obs[0], obs[1], noises[0], noises[1])
    h0 = ai.denoise_start(obs[0], 0, noises[0])
    h1 = ai.denoise_start(obs[1], 1, noises[1])
    assert h0 is not None and h1 is not None and h0["stream"].cuda_stream != h1["stream"].cuda_stream
    a, b = ai.denoise_join(h0), ai.denoise_join(h1)
    torch.cuda.synchronize()
    np.testing.assert_allclose(torch.cat([a, b]).cpu().numpy(), pair.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # replays reuse the chain's output buffer: the joined latents are copies
    h0b = ai.denoise_start(obs[1], 0, noises[1]); a2 = ai.denoise_join(h0b); torch.cuda.synchronize()
    np.testing.assert_allclose(a2.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert not torch.equal(a, a2)
    assert ai.denoise_start(obs[0][:64], 0, None) is None
    ai.concurrent_belief_loops = False
    assert ai.denoise_start(obs[0], 0, noises[0]) is None
