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
