"""DiffusionStateAgent (active_inference_diffusion/agents/state_agent.py:17-274): same constructor, `act`,
`train_step`, optimiser attributes and metric keys; each phase of train_step is one fused HIP call."""
from __future__ import annotations

import sys
from typing import Dict

import numpy as np
import torch

from .. import _native as nat
from .. import parallel
from ..core.active_inference import DiffusionActiveInference
from ..utils.buffers import ReplayBuffer
from .base_agent import BaseActiveInferenceAgent


def _is_box(space) -> bool:
    return hasattr(space, "shape") and space.shape is not None and len(space.shape) >= 1


class DiffusionStateAgent(BaseActiveInferenceAgent):
    def _setup_dimensions(self):
        self.obs_space, self.action_space = self.env.observation_space, self.env.action_space
        if not _is_box(self.obs_space):
            raise ValueError(f"Unsupported observation space: {type(self.obs_space)}")
        if not _is_box(self.action_space):
            raise ValueError(f"Unsupported action space: {type(self.action_space)}")
        self.observation_dim, self.action_dim = self.obs_space.shape[0], self.action_space.shape[0]
        self.config.observation_dim, self.config.action_dim = self.observation_dim, self.action_dim   # the reference mutates config too

    def _build_models(self):
        self.active_inference = DiffusionActiveInference(observation_dim=self.observation_dim, action_dim=self.action_dim,
                                                         latent_dim=self.config.latent_dim, config=self.config)

    def _create_replay_buffer(self) -> ReplayBuffer:
        return ReplayBuffer(capacity=self.training_config.buffer_size, obs_shape=(self.observation_dim,), action_dim=self.action_dim, device=self.device)

    def _process_observation(self, observation: np.ndarray) -> torch.Tensor:
        return torch.as_tensor(np.asarray(observation, dtype=np.float32)).unsqueeze(0)

    def _process_batch_observations(self, observations: torch.Tensor) -> torch.Tensor:
        return observations.to(self.device)

    overlap_streams = True    # False: every phase on the current stream, in order (used when timing individual kernels)

    def _side_workspace(self):
        if getattr(self, "_ws_side", None) is None:
            self._ws_side = nat.Workspace(self.device)
        return self._ws_side

    def _comm_stream(self):
        if getattr(self, "_comm", None) is None:
            # the process-wide side stream (aid_amd/streams.py: a process has four hardware queues, so every agent shares the same three extra streams)
            from ..streams import side_stream
            self._comm = side_stream(self.device)
            # everything this stream launches runs underneath the EFE's 25 600-row GEMMs: its small GEMMs take the 35 KB-LDS variant so
            # that they co-reside (a property of the stream, aid_stream_hint - no process-wide dispatch state)
            nat.stream_hint(self._comm, 1 if self.side_low_lds else 0)
        return self._comm

    timeline = None           # perf diagnostics (tools/perf/timeline.py): a list to which train_step appends (label, event) pairs

    def _mark(self, label, stream):
        if self.timeline is not None:
            import time as _t
            ev = torch.cuda.Event(enable_timing=True); ev.record(stream); self.timeline.append((label, ev, _t.perf_counter()))

    def enable_data_parallel(self):
        """Call once after torch.distributed.init_process_group: replicas start from rank 0's parameters."""
        for a in self.active_inference.arenas.values():
            parallel.broadcast_arena_(a.flat)
        self.active_inference.arenas["score"].reset_ema()
        self.active_inference.noise.seed += 7919 * parallel.rank()   # independent device-RNG streams per rank
        self.active_inference.param_version += 1
        self._dp_vote = {"phase": "warm", "n": 0, "t0": None, "ms": {}} if self.dp_autotune_schedule else None

    # -- data parallel: which belief-loop schedule to run -----------------------------------------------------------------------------
    # A process has four hardware queues (aid_amd/streams.py).  Main + side + two chain streams use all four; RCCL's internal stream is a
    # fifth, and a reverse loop that ends up sharing a queue with the estimator's stream costs 2x (98-108 ms against 54 per cfg-2 step).
    # With ONE rank on real RCCL the concurrent chains still paid (52.0 against 59.6 ms merged in round 3, tools/perf/dp_one_rank_time.py), but how
    # many streams an N-rank communicator uses is RCCL's business.  So a data-parallel agent MEASURES: after DP_VOTE_WARM warm-up steps it
    # times DP_VOTE_STEPS steps with the concurrent chains and as many with the merged loop (host clock around a device sync, each
    # rank; the slowest rank's time counts: one MAX all-reduce per arm), keeps the faster schedule for the rest of the run and says so
    # once.  Both schedules give the same results (tests/test_hip_agent.py::test_concurrent_belief_loops_equal_the_merged_loop), every rank
    # takes the same arm at the same step (the step counter of the vote is advanced identically on all ranks, the choice is made on the
    # all-reduced times), and nothing is re-executed.
    dp_autotune_schedule = True
    DP_VOTE_WARM, DP_VOTE_STEPS = 3, 3
    dp_schedule_choice = None          # "chains" | "merged" once decided (logged by bench.py)

    def _dp_schedule_vote(self) -> bool:
        """-> run this step's belief updates as concurrent chains?"""
        v = getattr(self, "_dp_vote", None)
        # the vote issues collectives (agree_max_float): whether it runs this step is the ranks' AGREED verdict (_take_prefetch), never a
        # rank-local one - a rank with concurrent_belief_loops / overlap_streams / dp_autotune_schedule switched differently would otherwise
        # pair its next collective with the other ranks' vote
        if v is None or not self._agreed.get("vote", False):
            return self.concurrent_belief_loops if self.dp_schedule_choice is None else self.dp_schedule_choice == "chains"
        import time as _t
        order = ("warm", "chains", "merged")
        limit = self.DP_VOTE_WARM if v["phase"] == "warm" else self.DP_VOTE_STEPS
        if v["n"] == limit:                                  # a phase has run its steps: close it, open the next
            torch.cuda.synchronize(self.device)
            now = _t.perf_counter()
            if v["phase"] != "warm":
                v["ms"][v["phase"]] = parallel.agree_max_float((now - v["t0"]) * 1e3 / limit, device=self.device)
            if v["phase"] == "merged":
                self.dp_schedule_choice = "chains" if v["ms"]["chains"] <= v["ms"]["merged"] else "merged"
                if parallel.rank() == 0:
                    print(f"[aid_amd] data parallel, {parallel.world()} ranks: belief loops as concurrent chains {v['ms']['chains']:.2f} ms/step, "
                          f"merged {v['ms']['merged']:.2f} ms/step (slowest rank) -> {self.dp_schedule_choice}", file=sys.stderr, flush=True)
                self.dp_schedule_ms = dict(v["ms"])
                self._dp_vote = None
                return self.dp_schedule_choice == "chains"
            v["phase"], v["n"], v["t0"] = order[order.index(v["phase"]) + 1], 0, _t.perf_counter()
        v["n"] += 1
        return v["phase"] != "merged"

    # ------------------------------------------------------------------------------------------------
    def train_step(self) -> Dict[str, float]:
        """One update of all four groups (state_agent.py:110-243).  Phases: minibatch gather -> reward normalisation ->
        belief latents for obs and next_obs -> ELBO step (score group) -> EFE step (policy) -> third belief update +
        lambda-returns + Huber (value) -> MSE (dynamics).  The epistemic-estimator training branch (`total_steps % 5 == 0`)
        cannot run in the reference and is skipped.  Metrics are read back from the device once, at the end.

        Scheduling (results unchanged): the reference's training loop runs hundreds of train_steps back to back on a fixed
        replay buffer (examples/train_mujoco.py:332-336).  The third belief update of step i and the two belief updates that open
        step i+1 use the SAME score parameters (the ones just updated), so step i draws step i+1's minibatch right after its own
        third-belief draws - the reference's draw order - and runs all three belief updates as ONE 3B-row reverse loop on a side
        stream underneath its EFE/MINE phase.  Step i+1 consumes the prefetched latents iff nothing they depend on has changed
        (replay-buffer version, score-parameter version, batch size, train mode); otherwise it recomputes them.  After a miss the
        prefetch backs off exponentially, so a train-once-per-env-step loop pays (almost) nothing for it."""
        cfg, ai = self.config, self.active_inference
        B = cfg.batch_size
        dp = parallel.is_dp()
        ready = len(self.replay_buffer) >= B
        if not ready and not dp:
            return {}
        main = torch.cuda.current_stream()
        pre = self._take_prefetch(B, ready)    # data parallel: a collective decision (one 8-int all-reduce), see _take_prefetch
        if not self._agreed["ready"]:          # some rank's replay shard is still short of a minibatch: no rank steps (the reference's `return {}`)
            self._lookahead = None
            return {}
        had_la = self._lookahead is not None
        la = self._take_lookahead(accepted=pre is not None)
        chains = self._dp_schedule_vote() if dp else self.concurrent_belief_loops
        ai.concurrent_belief_loops = self.overlap_streams and chains     # one stream only when asked for one (profiling)
        if had_la and not self.overlap_streams:
            # overlap_streams was switched off since the previous step: the look-ahead ELBO may still be running on the (real) side stream,
            # writing the score gradients and its shadow state, while everything below - the commit of that state and the score step if it
            # was accepted, zero_grad and this step's own ELBO if not - now runs on `main`
            main.wait_stream(self._comm_stream())
        elide = bool(self.elide_dead_next_belief)
        if pre is not None:
            batch, both = pre
        else:
            batch = self.replay_buffer.sample(B)
            if elide:
                # state_agent.py:138-139 computes a next_obs belief that :195-196 overwrites before anything reads it: only its DRAWS matter
                # (they advance the stream every later draw comes from).  A parity tape is advanced by exactly those draws; device Philox
                # has nothing to keep aligned.
                N_, tr_ = cfg.diffusion.num_diffusion_steps, ai.latent_score_network.training
                n_obs = ai.noise.belief(B, ai.latent_dim, cfg.hidden_dim, N_, tr_)
                ai.noise.belief(B, ai.latent_dim, cfg.hidden_dim, N_, tr_)          # the dead update's draws, consumed and dropped
                both = ai.denoise_only(batch["observations"], noise=n_obs, ws=ai._ws)      # ws given: the direct loop, as denoise_pair runs it
            else:
                both = ai.denoise_pair(batch["observations"], batch["next_observations"])
        obs, actions, rewards = batch["observations"], batch["actions"], batch["rewards"]
        next_obs, dones = batch["next_observations"], batch["dones"]
        if la is not None:          # this step's normalised rewards and ELBO were computed at the end of the previous step (_lookahead_elbo)
            nrew = la["nrew"]
        else:
            nrew = self.reward_normalizer.update_and_normalize(rewards, parallel.allgather_cat(rewards) if dp else None)
        latents = both[:B]
        # 2. score / diffusion group.  The policy phase reads no score-network parameter and no ELBO result, and the value phase needs a
        # fresh belief over next_obs from the UPDATED score network (state_agent.py:194-205).  So the whole chain
        #   ELBO forward + backward -> [gradient all-reduce ->] score step -> reverse loop
        # is enqueued on a side stream and runs underneath the EFE/MINE phase: its batch-256 kernels are latency-bound and leave most CUs
        # idle, the 25 600-row MINE GEMMs fill them.  Same arithmetic; the host enqueues ELBO, EFE, belief in the reference's order, so
        # the Philox offsets (draw order) are unchanged.
        side = self._comm_stream() if self.overlap_streams else main
        ws = self._side_workspace() if self.overlap_streams else None
        self._hint_main_stream(main, 2 if (self.overlap_streams and self.main_shares_cus) else 0)
        self._mark("start", main)
        if self.overlap_streams:
            side.wait_stream(main)
            for t_ in (obs, nrew, latents, next_obs):
                t_.record_stream(side)
        with torch.cuda.stream(side):
            if la is not None:
                # commit what the look-ahead computed on shadow copies (enqueued behind it on the same stream); the score gradients are in the arena
                self.reward_normalizer.state.copy_(la["rn_state"]); ai.time_importance_weights.copy_(la["tiw"]); ai._last_elbo = la["last_elbo"]
                elbo_info = la["info"]
            else:
                self.score_optimizer.zero_grad()
                elbo_loss, elbo_info = ai.compute_diffusion_elbo(obs, nrew, latents, ws=ws)
            self._mark("side: ELBO done", side)
            if not self.overlap_streams:
                self.score_optimizer.step()
            elif dp:
                # Issue the 109 MB gradient mean NOW, before the policy phase's collectives: a process group runs its collectives in
                # host order on one internal stream, so issued later it would queue behind the policy all-reduce (which waits for the
                # whole EFE phase) and the score step + reverse loop would lose their overlap with the EFE.  (Also for a look-ahead ELBO:
                # issued right behind it at the end of the previous step, the collective made the host wait for the look-ahead to finish
                # before the next step could start - one-rank RCCL run, tools/perf/dp_one_rank_time.py.)
                self.score_optimizer.allreduce_grads(async_op=True)
        # 3. policy
        self.policy_optimizer.zero_grad()
        efe, efe_info = ai.compute_expected_free_energy_diffusion(latents, horizon=cfg.efe_horizon, backward=True)
        self.policy_optimizer.step()
        self._mark("main: EFE + policy step done", main)
        # 4. value (targets from a re-sampled next-latent belief) [+ the two belief updates of the next step]
        with torch.cuda.stream(side):
            if self.overlap_streams:
                self.score_optimizer.step()            # data parallel: the 109 MB RCCL all-reduce runs here, off the critical path
            if self._want_prefetch():
                N, train = cfg.diffusion.num_diffusion_steps, ai.latent_score_network.training
                hook = lambda: ai.noise.belief(B, ai.latent_dim, cfg.hidden_dim, N, train)
                n3 = hook()                                        # draws of the third belief update ...
                nxt = self.replay_buffer.sample(B)                 # ... then the next step's index draw and its two belief updates
                if elide:       # the next step's next_obs belief is never read (see above): its draws are taken, the loop is not run
                    n1 = hook(); hook()
                    tri = ai.denoise_many([next_obs, nxt["observations"]], [n3, n1], ws=ws)
                else:
                    tri = ai.denoise_many([next_obs, nxt["observations"], nxt["next_observations"]], [n3, hook(), hook()], ws=ws)
                next_latents = tri[:B]
                self._prefetch = {"batch": nxt, "latents": tri[B:], "buffer": (id(self.replay_buffer), self.replay_buffer.version),
                                  "params": (self.score_optimizer.arena.step_count, ai.param_version), "B": B, "train": train}
            else:
                next_latents = ai.denoise_only(next_obs, ws=ws)
            self._mark("side: score step + reverse loop done", side)
        if self.overlap_streams:
            next_latents.record_stream(main); nrew.record_stream(main)
            for v in elbo_info.values():
                if torch.is_tensor(v):
                    v.record_stream(main)
            if self._prefetch is not None:
                self._prefetch["latents"].record_stream(main)
                for v in self._prefetch["batch"].values():
                    v.record_stream(main)
            main.wait_stream(side)
        self.value_optimizer.zero_grad()
        nl_x, nrew_x, dones_x = parallel.halo_from_next_rank(next_latents, nrew, dones, rows=5) if dp else (next_latents, nrew, dones)
        value_loss, _ = nat.value_step(ai.value_network.table(), ai.value_network.grad_table(), latents, nl_x, nrew_x, dones_x,
                                       cfg.discount_factor, 0.95, 5, ws=ai._ws)
        self.value_optimizer.step()
        epi_metrics = {}
        if self.train_epistemic_every and self.total_steps % self.train_epistemic_every == 0:   # state_agent.py:216-222
            epistemic_mi, epi_metrics = ai.train_epistemic_estimator(latents, actions, next_latents)
            epi_metrics = {"epistemic_mi": epistemic_mi, **epi_metrics}
        # 5. dynamics
        self.dynamics_optimizer.zero_grad()
        dyn_loss = nat.dynamics_step(ai.latent_dynamics.table(), ai.latent_dynamics.grad_table(), latents, actions, next_latents, ws=ai._ws)
        self.dynamics_optimizer.step()
        self._mark("main: value + dynamics done", main)
        if self._agreed["lookahead"] and self.overlap_streams and self._prefetch is not None:
            self._lookahead_elbo(side, ws, dp, main)
        self.total_steps += 1
        # one device->host read for every metric
        keys, vals = [], []
        for k, v in {**elbo_info, "policy_loss": efe_info.pop("policy_loss"), **efe_info, "value_loss": value_loss, "dynamics_loss": dyn_loss}.items():
            if torch.is_tensor(v):
                keys.append(k); vals.append(v.reshape(1))
        host = torch.cat(vals).cpu().tolist()
        metrics = dict(zip(keys, host))
        metrics.update({"num_trajectories": efe_info["num_trajectories"], "horizon": efe_info["horizon"]})
        metrics.update(epi_metrics)
        self._last = {"latents": latents, "next_latents": next_latents}
        return metrics

    # The reference trains the MINE estimator when `total_steps % 5 == 0` (state_agent.py:216-222) but its concrete agents never create
    # the optimiser that branch needs, so it raises there.  Opt in with 5 to get the behaviour the code intends.
    train_epistemic_every = 0
    side_low_lds = True            # side-stream launches (ELBO, score step, reverse loop) use the 35 KB panel variant

    # -- cross-step prefetch of the next minibatch's belief latents ------------------------------------------------------------
    prefetch_next_batch = True     # False: never look ahead (every train_step samples and denoises its own minibatch)
    _prefetch = None
    _prefetch_backoff = 0          # train_steps to wait before trying again after an unused prefetch
    _prefetch_wait = 0

    # -- look-ahead ELBO ----------------------------------------------------------------------------------------------------------
    # With the next minibatch and its belief latents prefetched, everything the NEXT step's ELBO reads exists once this step's score update has
    # run: the score parameters (no other update touches them in between), the minibatch, its latents, and the reward normaliser (whose next
    # update is that minibatch's).  So the next step's ELBO forward + backward is enqueued on the side stream at the END of this step - where
    # it runs next to the value / dynamics updates and the next step's rollout, all of which share CUs with it - instead of at the start of the
    # next step underneath the estimator's 1024-wide kernel, which shares a CU with nothing: there its ~400 small dependent kernels queue behind
    # 300-us workgroups (24 ms for 7.4 ms of work, tools/perf/timeline.py) and the reverse loop that follows them starts that much later.
    # Results are those of the reference order: the draws are made in the same sequence (..., third belief, next indices, its two beliefs,
    # its ELBO, ...), the two pieces of state the ELBO advances - reward normaliser, time-importance weights - are advanced on shadow copies and
    # committed when the next step accepts the prefetch (same validity test), and discarded otherwise.
    lookahead_elbo = True
    _lookahead = None
    # The reference computes a belief over next_obs at the top of the step (state_agent.py:138-139) and overwrites it with a re-sampled one
    # (:195-196) before anything reads it - the pixel agent does read its (pixel_agent.py:364-371), the state agent does not.  True: that
    # reverse loop is not run (one of the three 50-step loops of a step; its random draws are still consumed from a parity tape, so every
    # later draw, every metric and every post-step tensor is unchanged: tests/test_hip_agent.py::test_dead_next_belief_elision_changes_nothing).
    # False: executed as the reference executes it.
    elide_dead_next_belief = True
    concurrent_belief_loops = True     # the belief updates of a step as separate concurrent reverse loops (DiffusionActiveInference._denoise_chains)

    def _lookahead_elbo(self, side, ws, dp, main):
        ai, pre = self.active_inference, self._prefetch
        nb, B = pre["batch"], pre["B"]
        with torch.cuda.stream(side):
            rew = nb["rewards"].contiguous().float()
            rn = self.reward_normalizer.state.clone()
            nrew = nat.reward_normalize_global(parallel.allgather_cat(rew), rew, rn) if dp else nat.reward_normalize(rew, rn)
            tiw_live, last_live = ai.time_importance_weights, getattr(ai, "_last_elbo", None)
            tiw = tiw_live.clone()
            ai.time_importance_weights = tiw
            try:
                self.score_optimizer.zero_grad()
                _, info = ai.compute_diffusion_elbo(nb["observations"], nrew, pre["latents"][:B], ws=ws)
            finally:
                ai.time_importance_weights = tiw_live
                last, ai._last_elbo = ai._last_elbo, last_live
            self._mark("side: look-ahead ELBO of the next step done", side)
        self._lookahead = {"pre": pre, "nrew": nrew, "rn_state": rn, "tiw": tiw, "info": info, "last_elbo": last}

    def _take_lookahead(self, accepted: bool):
        """The look-ahead ELBO of this step if its prefetch was accepted; otherwise it is dropped (its shadow state is never committed, its
        gradients are overwritten by this step's own ELBO)."""
        la, self._lookahead = self._lookahead, None
        if la is None:
            return None
        return la if accepted else None

    # What this step does about prefetching, as agreed by all ranks (one rank: its own switches).  Written by _take_prefetch at the start of
    # every train_step and read by everything further down that gates a collective.
    _agreed = {"prefetch": True, "lookahead": True, "vote": False, "ready": True}

    def _want_prefetch(self) -> bool:
        self._prefetch = None
        if not self._agreed["prefetch"]:
            return False
        if self._prefetch_wait > 0:
            self._prefetch_wait -= 1
            return False
        return True

    def _take_prefetch(self, B: int, ready: bool = True):
        """The prefetched (minibatch, latents) if they are exactly what this step would compute, else None.

        Data parallel: accepting or rejecting changes which collectives the step issues (an accepted look-ahead ELBO has already run its
        all-gathers; a rejected one makes the step run them again before the score all-reduce; the look-ahead itself issues three more), so
        the decision must not be rank-local: a rank whose replay buffer was touched between two steps would pair its all-gather with the
        other ranks' all-reduce.  Every rank therefore contributes its local verdict AND the state of its prefetch state machine (back-off
        counters, the two switches) to ONE element-wise MIN over ranks (`parallel.agree_min`) and all of them act on the result: accept only
        if every rank would, wait as long as the most cautious rank, prefetch / look ahead only if every rank has it switched on.  After
        the call the state machines are identical on all ranks, so `_want_prefetch` and the look-ahead further down need no second exchange."""
        pre, self._prefetch = self._prefetch, None
        ai = self.active_inference
        ok = (pre is not None and pre["buffer"] == (id(self.replay_buffer), self.replay_buffer.version) and pre["B"] == B and
              pre["params"] == (self.score_optimizer.arena.step_count, ai.param_version) and pre["train"] == ai.latent_score_network.training)
        had = pre is not None
        want_pre, want_la = bool(self.prefetch_next_batch), bool(self.lookahead_elbo and self.overlap_streams)
        # the schedule vote (_dp_schedule_vote) and the "buffer still short of a minibatch" early return both decide whether collectives are
        # issued further down: they ride on the same exchange
        vote = bool(getattr(self, "_dp_vote", None) is not None and self.concurrent_belief_loops and self.overlap_streams)
        if parallel.is_dp():
            v = parallel.agree_min([ok, -int(had), -self._prefetch_wait, -self._prefetch_backoff, want_pre, want_la, vote, ready], device=self.device)
            ok, had, self._prefetch_wait, self._prefetch_backoff, want_pre, want_la = bool(v[0]), bool(-v[1]), -v[2], -v[3], bool(v[4]), bool(v[5])
            vote, ready = bool(v[6]), bool(v[7])
        self._agreed = {"prefetch": want_pre, "lookahead": want_la, "vote": vote, "ready": ready}
        if not ready:
            return None
        if ok:
            self._prefetch_backoff = 0
            return pre["batch"], pre["latents"]
        if had:         # a prefetch that (some rank) could not use: back off
            self._prefetch_backoff = min(64, max(1, 2 * self._prefetch_backoff))
            self._prefetch_wait = self._prefetch_backoff
        return None

    def invalidate_prefetch(self):
        """Call after changing score-network parameters by hand (load_state_dict and the optimisers do it themselves)."""
        self._prefetch = None
