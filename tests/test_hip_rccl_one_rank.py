"""GPU: the data-parallel train_step over the REAL "nccl" (= RCCL) backend with one rank.

The 1-GPU test box cannot host two RCCL ranks (one GPU per rank), so the multi-rank tests run over gloo.  What gloo cannot show is that
the calls themselves are accepted by RCCL on this hardware: `ReduceOp.AVG`, `all_gather_into_tensor` on fp32 / fp64, `broadcast`, an
asynchronous all-reduce issued on the side stream and waited for on it, `init_process_group("nccl", device_id=...)`.  Here a single-rank
process group runs the agent with `parallel.is_dp()` forced on, so every collective of the DP path executes on RCCL (degenerate: a mean
over one rank is the identity), and two train_steps must give exactly what the non-distributed agent gives on the same noise tape.
Runs in a child process: a process group must not leak into the test runner."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

O, A, D, N, H, B, CAP = 17, 6, 64, 4, 2, 8, 64


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(dp, pixel):
    import aid_amd
    from oracle.noise import NoiseStream
    from parity_noise import ParityNoise

    class Box:
        def __init__(self, shape): self.shape = shape
    torch.manual_seed(11)
    if pixel:
        L, Ap, Bp = 32, 4, 4
        cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=Bp, efe_horizon=2, latent_dim=L, hidden_dim=64, pixel_observation=True)
        cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=3)
        class Env: observation_space, action_space = Box((3, 3, 84, 84)), Box((Ap,))
        agent = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=16), aid_amd.PixelObservationConfig())
        rng = np.random.default_rng(3)
        for _ in range(8):
            agent.replay_buffer.add(rng.random((3, 3, 84, 84), dtype=np.float32), rng.uniform(-1, 1, Ap), float(rng.standard_normal()),
                                    rng.random((3, 3, 84, 84), dtype=np.float32), False)
        pn = ParityNoise(99, pixel=True); agent.active_inference.noise = pn; agent.replay_buffer.index_source = pn      # one explicit tape on both sides
    else:
        cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=H, latent_dim=O, hidden_dim=D)
        cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)
        class Env: observation_space, action_space = Box((O,)), Box((A,))
        agent = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=CAP))
        bs = NoiseStream(5)
        agent.replay_buffer.add_batch(bs.randn(CAP, O), np.tanh(bs.randn(CAP, A)).astype(np.float32), bs.randn(CAP), bs.randn(CAP, O), bs.rand(CAP) < 0.15)
        pn = ParityNoise(99); agent.active_inference.noise = pn; agent.replay_buffer.index_source = pn
    if dp:
        agent.enable_data_parallel()
    ms = [agent.train_step() for _ in range(2)]
    torch.cuda.synchronize()
    arenas = dict(agent.active_inference.arenas)
    if pixel:
        arenas["encoder"] = agent.encoder_arena
    return ms, {k: a.flat.detach().cpu().numpy().copy() for k, a in arenas.items()}, agent.reward_normalizer.state.cpu().numpy().copy()


def _worker(port, q, pixel):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        ref = _run(False, pixel)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        from aid_amd import parallel
        assert dist.get_backend() == "nccl"
        parallel.is_dp = lambda: True                     # one rank, but take every data-parallel branch
        # the wrappers on their own: AVG all-reduce (sync and async on a side stream), all-gather of fp32 / fp64, broadcast
        x = torch.arange(1024, device="cuda", dtype=torch.float32); x0 = x.clone()
        parallel.allreduce_mean_(x); assert torch.equal(x, x0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            h = parallel.allreduce_mean_(x, async_op=True); h.wait()
        torch.cuda.current_stream().wait_stream(side); assert torch.equal(x, x0)
        for dt in (torch.float32, torch.float64):
            y = torch.randn(8, 3, device="cuda").to(dt)
            assert torch.equal(parallel.allgather_cat(y), y)
        parallel.broadcast_arena_(x); assert torch.equal(x, x0)
        z, = parallel.halo_from_next_rank(torch.randn(8, 4, device="cuda")); assert z.shape == (8, 4)      # the last rank appends nothing
        got = _run(True, pixel)
        torch.cuda.synchronize()
        for step in range(2):
            for k, v in ref[0][step].items():
                assert got[0][step][k] == v or (np.isnan(v) and np.isnan(got[0][step][k])), (step, k, got[0][step][k], v)
        for k in ref[1]:
            np.testing.assert_array_equal(got[1][k], ref[1][k], err_msg=k)
        np.testing.assert_array_equal(got[2], ref[2])
        q.put("ok")
    except Exception:
        import traceback
        q.put(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("pixel", [False, True])
def test_dp_train_step_on_real_rccl_with_one_rank(pixel):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q, pixel))
    p.start()
    msg = q.get(timeout=600)
    p.join(60)
    assert msg == "ok", msg
