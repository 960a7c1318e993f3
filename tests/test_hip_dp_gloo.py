"""GPU, 2 ranks sharing cuda:0 over gloo: the agent's data-parallel train_step end to end (broadcast of the arenas, global reward
normaliser, rank-major time-importance update, gradient means on the side stream, per-rank Philox streams).  RCCL needs one GPU per
rank, which the 1-GPU test box cannot offer; gloo moves the same tensors through the same aid_amd.parallel calls."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _replicas_identical(agent, rank):
    torch.cuda.synchronize()
    for name, a in agent.active_inference.arenas.items():
        mine = a.flat.detach().clone()
        other = mine.clone(); dist.broadcast(other, src=0)
        assert torch.equal(mine, other), f"arena {name} diverged on rank {rank}"
    for x in (agent.reward_normalizer.state, agent.active_inference.time_importance_weights):
        mine = x.clone(); other = mine.clone(); dist.broadcast(other, src=0)
        assert torch.equal(mine, other)


def _scenario_worker(rank, world, port, q, scenario):
    """`diverge`: the cross-step prefetch + look-ahead ELBO under data parallelism when ONE rank's replay buffer changes between two steps
    (an asynchronous collector adding a transition): the accept decision is collective, so both ranks drop the prefetch and the look-ahead
    together, back off together and pick the prefetch up again together - no hang, replicas bit-identical.
    `chains`: more than 128 rows per rank, so the belief updates run as hipGraph-replayed concurrent chains under DP, through the
    start-up vote between the chain and the merged schedule (one step each here); every rank lands on the same choice."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import aid_amd
        from aid_amd import parallel

        class Box:
            def __init__(self, shape): self.shape = shape
        torch.manual_seed(200 + rank)
        rng = np.random.default_rng(17 + rank)
        O, A = 17, 6
        B, H, cap = (8, 2, 64) if scenario == "diverge" else (136, 1, 512)
        cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=H, latent_dim=O, hidden_dim=64)
        cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=4)
        class Env: observation_space, action_space = Box((O,)), Box((A,))
        agent = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=cap))
        agent.replay_buffer.add_batch(rng.standard_normal((cap // 2, O)).astype(np.float32), rng.uniform(-1, 1, (cap // 2, A)).astype(np.float32),
                                      rng.standard_normal(cap // 2).astype(np.float32), rng.standard_normal((cap // 2, O)).astype(np.float32),
                                      np.zeros(cap // 2, bool))
        agent.DP_VOTE_WARM, agent.DP_VOTE_STEPS = 1, 1
        agent.enable_data_parallel()
        assert agent.prefetch_next_batch and agent.lookahead_elbo
        if scenario == "diverge":
            for step in range(6):
                if step == 3 and rank == 1:
                    agent.replay_buffer.add(rng.standard_normal(O), rng.uniform(-1, 1, A), 0.5, rng.standard_normal(O), False)
                m = agent.train_step()
                assert all(np.isfinite(v) for v in m.values()), (step, m)
                if step == 3:      # rank 0 alone would have accepted; both dropped it and sit out one step
                    assert agent._prefetch is None and agent._lookahead is None and agent._prefetch_backoff == 1, (rank, step)
                else:
                    assert agent._prefetch is not None and agent._lookahead is not None, (rank, step)
        else:
            for step in range(5):
                m = agent.train_step()
                assert all(np.isfinite(v) for v in m.values()), (step, m)
            assert agent.dp_schedule_choice in ("chains", "merged") and agent._dp_vote is None
            # the chain arm ran from captured graphs: one per live belief update of a step (two with the dead next_obs update elided, else three)
            assert len(agent.active_inference.__dict__.get("_chain_graphs", {})) >= (2 if agent.elide_dead_next_belief else 3)
            choice = parallel.agree_min([agent.dp_schedule_choice == "chains"]) + parallel.agree_min([-(agent.dp_schedule_choice == "chains")])
            assert choice[0] == -choice[1]
        _replicas_identical(agent, rank)
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _worker(rank, world, port, q, pixel):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import aid_amd
        from aid_amd import parallel

        class Box:
            def __init__(self, shape): self.shape = shape
        torch.manual_seed(100 + rank)                    # different initial weights per rank: enable_data_parallel must reconcile them
        rng = np.random.default_rng(7 + rank)            # different replay shards per rank
        if pixel:
            L, A, B = 32, 4, 4
            cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=2, latent_dim=L, hidden_dim=64, pixel_observation=True)
            cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=3)
            class Env: observation_space, action_space = Box((3, 3, 84, 84)), Box((A,))
            agent = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=16), aid_amd.PixelObservationConfig())
            for _ in range(8):
                agent.replay_buffer.add(rng.random((3, 3, 84, 84), dtype=np.float32), rng.uniform(-1, 1, A), float(rng.standard_normal()),
                                        rng.random((3, 3, 84, 84), dtype=np.float32), False)
        else:
            O, A, B = 17, 6, 8
            cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=2, latent_dim=O, hidden_dim=64)
            cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=4)
            class Env: observation_space, action_space = Box((O,)), Box((A,))
            agent = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=64))
            for _ in range(32):
                agent.replay_buffer.add(rng.standard_normal(O), rng.uniform(-1, 1, A), float(rng.standard_normal()), rng.standard_normal(O), False)
        agent.enable_data_parallel()
        assert parallel.is_dp() and parallel.world() == world
        for _ in range(3):
            m = agent.train_step()
        assert all(np.isfinite(v) for v in m.values()), m
        torch.cuda.synchronize()
        # replicas stay bit-identical: same averaged gradients, same optimiser state
        arenas = dict(agent.active_inference.arenas)
        if pixel:
            arenas["encoder"] = agent.encoder_arena
        for name, a in arenas.items():
            mine = a.flat.detach().clone()
            other = mine.clone(); dist.broadcast(other, src=0)
            assert torch.equal(mine, other), f"arena {name} diverged on rank {rank}"
        # the reward normaliser saw the global batch: identical statistics everywhere
        st = agent.reward_normalizer.state.clone(); st0 = st.clone(); dist.broadcast(st0, src=0)
        assert torch.equal(st, st0)
        assert float(agent.reward_normalizer.count) > 3 * world * cfg.batch_size - 1
        tiw = agent.active_inference.time_importance_weights.clone(); t0 = tiw.clone(); dist.broadcast(t0, src=0)
        assert torch.equal(tiw, t0)
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("pixel", [False, True])
def test_agent_data_parallel_two_ranks(pixel):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, pixel)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs: p.join(60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}:\n{msg}"


@pytest.mark.parametrize("scenario", ["diverge", "chains"])
def test_dp_prefetch_decisions_are_collective(scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_scenario_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs: p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]          # a mismatched collective shows up here as a timeout, not a hang of the suite
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    for r, msg in res:
        assert msg == "ok", f"rank {r}:\n{msg}"


def test_bench_strong_scaling_rehearsal_four_ranks_on_one_gpu():
    """`bench.py --gpus 4 --scaling strong` end to end with the four ranks sharing cuda:0 over gloo (AID_BENCH_REHEARSAL=gloo): the launcher, the
    256-row global batch split into 64-row shards, every data-parallel collective of the real train_step at that size, both timed regions
    (dead next_obs belief executed / elided) and the JSON line with the ranks the communicator really had.  The number is not a result (one GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AID_BENCH_REHEARSAL="gloo", AID_BENCH_SKIP="phases,graphs", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--scaling", "strong", "--steps", "2", "--warmup", "2", "--no-extras",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["config"]["global_batch"] == 256 and out["config"]["rows_per_rank"] == 64
    dp = out["data_parallel"]
    assert dp["rccl_ranks_seen"] == 4 and dp["backend"] == "gloo" and len(dp["per_rank_ms_per_step"]) == 4 and len(dp["per_rank_ms_per_step_dead_belief_elided"]) == 4
    assert out["value"] > 0 and out["value_dead_belief_elided"] > 0 and "REHEARSAL" in out["data"]
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) < 1e-5 * out["value"]          # strong: ONE global batch-256 step per step time (the line carries 6 significant digits)
