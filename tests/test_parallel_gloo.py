"""CPU, world_size 2, gloo: the data-parallel host logic (aid_amd/parallel.py) and the equivalence it relies on —
averaging per-shard gradients of mean-losses equals the gradient of the global batch (checked with the oracle)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _prefetch_state_machine(rank):
    """DiffusionStateAgent's accept / back-off logic for the cross-step prefetch is collective under data parallelism (every collective of
    the look-ahead ELBO hangs off it): driven here on a bare instance (no GPU: only the host-side state machine) through the sequence
      step 0: both ranks hold a valid prefetch                    -> both accept
      step 1: rank 1's replay buffer changed in between            -> BOTH reject and back off one step
      step 2: back-off                                             -> neither prefetches
      step 3: rank 0 has prefetching switched off                  -> neither prefetches nor looks ahead
    and after every step the two state machines must be identical."""
    from types import SimpleNamespace as NS
    from aid_amd.agents.state_agent import DiffusionStateAgent
    ag = object.__new__(DiffusionStateAgent)
    ag.device = "cpu"
    ag.replay_buffer = NS(version=3)
    ag.score_optimizer = NS(arena=NS(step_count=7))
    ag.active_inference = NS(param_version=1, latent_score_network=NS(training=True))
    mk = lambda: {"batch": "b", "latents": "l", "buffer": (id(ag.replay_buffer), ag.replay_buffer.version), "params": (7, 1), "B": 8, "train": True}
    state = lambda: (ag._prefetch_wait, ag._prefetch_backoff, ag._agreed["prefetch"], ag._agreed["lookahead"])
    def same_everywhere(x):
        v = [int(a) for a in x]
        lo, hi = parallel.agree_min(v), [-a for a in parallel.agree_min([-a for a in v])]
        assert lo == hi, (x, lo, hi)
    from aid_amd import parallel
    ag._prefetch = mk()
    assert ag._take_prefetch(8) == ("b", "l") and ag._want_prefetch(); same_everywhere(state())
    ag._prefetch = mk()
    if rank == 1:
        ag.replay_buffer.version += 1
    assert ag._take_prefetch(8) is None                      # rank 0 alone would have accepted
    assert (ag._prefetch_wait, ag._prefetch_backoff) == (1, 1); same_everywhere(state())
    assert not ag._want_prefetch()                           # the back-off step, on both ranks
    assert ag._take_prefetch(8) is None and ag._prefetch_backoff == 1      # nothing was prefetched anywhere: no further back-off
    assert ag._want_prefetch(); same_everywhere(state())
    if rank == 0:
        ag.prefetch_next_batch = False
    ag._prefetch = mk()
    ag._take_prefetch(8)
    assert not ag._agreed["prefetch"] and not ag._want_prefetch(); same_everywhere(state())
    ag.prefetch_next_batch, ag.overlap_streams = True, rank == 0          # rank 1 runs single-stream: no look-ahead anywhere
    ag._take_prefetch(8)
    assert ag._agreed == {"prefetch": True, "lookahead": False, "vote": False, "ready": True}
    # the schedule vote and the "replay shard still short of a minibatch" early return also gate collectives: both follow the agreed verdict.
    # rank 1 has the vote pending but rank 0 has concurrent chains switched off -> nobody votes (no agree_max_float is issued: a mismatched
    # collective here would hang this test); rank 0's buffer is short -> nobody steps
    ag.overlap_streams = True
    ag._dp_vote = {"phase": "warm", "n": 0, "t0": None, "ms": {}}
    ag.concurrent_belief_loops = rank == 1
    ag._take_prefetch(8, ready=True)
    assert ag._agreed["vote"] is False and ag._agreed["ready"] is True
    assert ag._dp_schedule_vote() == ag.concurrent_belief_loops and ag._dp_vote["n"] == 0        # untouched: the vote did not run
    ag.concurrent_belief_loops = True
    ag._take_prefetch(8, ready=rank == 1)
    assert ag._agreed["vote"] is True and ag._agreed["ready"] is False
    ag._take_prefetch(8, ready=True)
    assert ag._agreed["ready"] is True and ag._dp_schedule_vote() is True and ag._dp_vote["n"] == 1    # both ranks advance the vote together


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from aid_amd import parallel
        from oracle import aid_oracle as orc
        from oracle.weights import draw_params, param_specs
        assert parallel.is_dp() and parallel.world() == world and parallel.rank() == rank
        # all-reduce mean of a flat arena
        g = torch.full((1000,), float(rank + 1))
        parallel.allreduce_mean_(g)
        assert torch.allclose(g, torch.full((1000,), 1.5))
        # rank-major all-gather, broadcast, disjoint index shards
        x = torch.arange(4, dtype=torch.float32) + 10 * rank
        assert parallel.allgather_cat(x).tolist() == [0, 1, 2, 3, 10, 11, 12, 13]
        w = torch.full((8,), float(rank)); parallel.broadcast_arena_(w); assert float(w.sum()) == 0.0
        idx = np.arange(16); mine = parallel.shard_indices(idx)
        assert mine.tolist() == list(range(8 * rank, 8 * rank + 8))
        with pytest.raises(ValueError):
            parallel.shard_indices(np.arange(15))
        # DP equivalence on the dynamics + value losses (mean over the batch): grad(global) == mean_r grad(shard_r)
        O, A, D, N, B = 11, 3, 32, 4, 8
        p = draw_params(param_specs(O, A, O, D, N), 5, orc.schedule_tables(N))
        rng = np.random.default_rng(0)
        z, a, zn = (torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((B, O), (B, A), (B, O)))
        names = orc.group_names(p, "dynamics")
        for n in names: p[n].requires_grad_(True)
        def loss(sl):
            pred, _ = orc.predict_next_latent(p, z[sl], a[sl]); return torch.nn.functional.mse_loss(pred, zn[sl])
        full = torch.autograd.grad(loss(slice(0, B)), [p[n] for n in names])
        half = B // world
        mine = torch.autograd.grad(loss(slice(rank * half, (rank + 1) * half)), [p[n] for n in names])
        flat = torch.cat([t.reshape(-1) for t in mine]); parallel.allreduce_mean_(flat)
        assert torch.allclose(flat, torch.cat([t.reshape(-1) for t in full]), rtol=1e-5, atol=1e-7)
        # global reward-normaliser moments == moments of the gathered batch
        r = torch.from_numpy(rng.standard_normal(B).astype(np.float32))
        allr = parallel.allgather_cat(r[rank * half:(rank + 1) * half].contiguous())
        assert torch.equal(allr, r)
        # collective decisions: element-wise MIN of host integers, MAX of a host float
        assert parallel.agree_min([1, -rank, 5 + rank]) == [1, -1, 5]
        assert parallel.agree_min([rank == 0]) == [0]
        assert parallel.agree_max_float(1.5 + rank) == 2.5
        _prefetch_state_machine(rank)
        q.put((rank, "ok"))
    except Exception as e:   # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_host_logic_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(v == "ok" for _, v in res), res
