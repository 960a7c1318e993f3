"""DiffusionPixelAgent (active_inference_diffusion/agents/pixel_agent.py:19-470): DrQ-v2 encoder -> features -> the same
diffusion active-inference core with observation_dim = latent_dim.  Same constructor, `act`, `encode_observation`,
`train_step`, `compute_representation_loss`, optimiser attributes and metric keys as the reference.

The ConvDecoder and the pixel branch of the epistemic estimator are built for inference (`aid_conv_decoder_forward`,
`aid_pixel_epistemic_value`).  The reference evaluates that estimator inside every EFE call under no_grad (about 40 GFLOP per sample
row, 25 600 rows per horizon step at K = S = 10, B = 256); it changes no parameter update, so here it is opt-in:
`agent.active_inference.pixel_epistemic = True` adds the epistemic term, its metrics and the MINE running mean exactly as the reference."""
from __future__ import annotations

from typing import Any, Dict, Tuple

import numpy as np
import torch

from .. import _native as nat
from .. import parallel
from ..arena import ArenaOptimizer, ParamArena, adamw_load_state_dict, adamw_state_dict
from ..configs.config import PixelObservationConfig
from ..core.active_inference import DiffusionActiveInference
from ..encoder.visual_encoders import DrQV2Encoder, RandomShiftAugmentation
from ..utils.buffers import ReplayBuffer
from .base_agent import BaseActiveInferenceAgent


class _PixelScoreOptimizer:
    """The reference's single AdamW over score net + diffusion + encoder + feature decoder (pixel_agent.py:441-449) with
    its clip over the first three only (:325-331): three arenas, one shared squared-norm."""

    def __init__(self, score: ParamArena, encoder: ParamArena, fdec: ParamArena, lr: float, clip: float, layout=None):
        self.score, self.encoder, self.fdec, self.clip = score, encoder, fdec, clip
        self.layout = layout
        self.param_groups = [{"lr": lr, "weight_decay": 1e-5, "betas": (0.9, 0.999), "eps": 1e-8}]
        self._pending = None

    def zero_grad(self, set_to_none: bool = False):
        for a in (self.score, self.encoder, self.fdec):
            a.zero_grad()

    def allreduce_grads(self, async_op: bool = False):
        self._pending = [parallel.allreduce_mean_(a.grad, async_op=async_op) for a in (self.score, self.encoder, self.fdec)]

    HEAD = "encoder.output_layers.0.weight"       # Linear(451 584 -> 2L): 462 MB of the 468 MB encoder arena at L = 128

    def allreduce_encoder_head(self):
        """Data parallel: start the mean of the Linear(451 584 -> 2L) gradient - the first thing the encoder backward produces (its head phase) - now, so that it
        travels underneath the conv stack's backward; step() then reduces only what is left of the encoder arena.  Issued on the calling (side) stream."""
        o, shape = self.encoder.offsets[self.HEAD]
        n = int(torch.Size(shape).numel())
        self._head = (o, n, parallel.allreduce_mean_(self.encoder.grad[o:o + n], async_op=True))

    def step(self):
        if self._pending is not None:
            for h in self._pending:
                if h is not None:
                    h.wait()
            self._pending = None
        else:
            head, self._head = getattr(self, "_head", None), None
            parallel.allreduce_mean_(self.score.grad)
            if head is None:
                parallel.allreduce_mean_(self.encoder.grad)
            else:       # the head slice is already on its way: the rest of the arena in (at most) two pieces around it, then wait for the head
                o, n, h = head
                if o > 0:
                    parallel.allreduce_mean_(self.encoder.grad[:o])
                if o + n < self.encoder.grad.numel():
                    parallel.allreduce_mean_(self.encoder.grad[o + n:])
                if h is not None:
                    h.wait()
            parallel.allreduce_mean_(self.fdec.grad)
        g = self.param_groups[0]
        ss, se = self.score.grad_sumsq().clone(), self.encoder.grad_sumsq().clone()
        self.score.step(g["lr"], g["weight_decay"], self.clip, 0.9999, ext_sumsq=se)
        self.encoder.step(g["lr"], g["weight_decay"], self.clip, ext_sumsq=ss)
        self.fdec.step(g["lr"], g["weight_decay"], 3.0e38)       # in the optimiser, outside the clipped set

    def state_dict(self):
        """One torch.optim.AdamW state over score net + diffusion + encoder + feature decoder, by position in the reference's list."""
        return adamw_state_dict(self.layout, self.param_groups[0])

    def load_state_dict(self, sd):
        adamw_load_state_dict(self.layout, self.param_groups[0], sd)


class DiffusionPixelAgent(BaseActiveInferenceAgent):
    def __init__(self, env, config, training_config, pixel_config: PixelObservationConfig):
        self.pixel_config = pixel_config
        self.frame_stack = 1
        super().__init__(env, config, training_config)

    def _setup_dimensions(self):
        self.obs_space, self.action_space = self.env.observation_space, self.env.action_space
        shape = getattr(self.obs_space, "shape", None)
        if shape is None:
            raise ValueError(f"Unsupported observation space: {type(self.obs_space)}")
        if len(shape) == 3:      # (C, H, W), or (H, W, C) when the last axis looks like channels
            self.obs_shape = (shape[-1], shape[0], shape[1]) if shape[-1] in (1, 3) else tuple(shape)
        elif len(shape) == 4:
            self.frame_stack, self.obs_shape = shape[0], tuple(shape[1:])
        else:
            raise ValueError(f"Unexpected observation shape: {shape}")
        if getattr(self.action_space, "shape", None) is None:
            raise ValueError(f"Unsupported action space: {type(self.action_space)}")
        self.action_dim = self.action_space.shape[0]
        self.config.action_dim = self.action_dim
        self.config.observation_dim = self.config.latent_dim

    def _build_models(self):
        pc = self.pixel_config
        self.encoder = DrQV2Encoder(obs_shape=self.obs_shape, feature_dim=self.config.latent_dim, frame_stack=pc.frame_stack, num_layers=4, num_filters=32)
        self.augmentation = RandomShiftAugmentation(pad=pc.random_shift_pad) if pc.augmentation else None
        self.active_inference = DiffusionActiveInference(observation_dim=self.config.observation_dim, action_dim=self.action_dim,
                                                         latent_dim=self.config.latent_dim, config=self.config, pixel_shape=self.obs_shape)
        self.encoder.to(self.device)
        self.encoder_arena = ParamArena([("encoder." + n, p) for n, p in self.encoder.trained_parameters()], self.device)
        self.encoder.rng_seed = self.active_inference.noise.seed ^ 0x5EED

    def _setup_optimizers(self):
        a, lr, clip = self.active_inference.arenas, self.config.learning_rate, self.config.gradient_clip
        enc_layout = [(self.encoder_arena, "encoder." + n) if ("encoder." + n) in self.encoder_arena.offsets else None   # channel_reduce: listed, never stepped
                      for n, _ in self.encoder.named_parameters()]
        fd_layout = [(a["feature_decoder"], "feature_decoder." + n) for n, _ in self.active_inference.feature_decoder.named_parameters()]
        self.score_optimizer = _PixelScoreOptimizer(a["score"], self.encoder_arena, a["feature_decoder"], lr, clip,
                                                    layout=self._score_layout() + enc_layout + fd_layout)
        self.policy_optimizer = ArenaOptimizer(a["policy"], lr, 1e-5, clip)
        self.value_optimizer = ArenaOptimizer(a["value"], lr, 1e-2, clip)
        self.dynamics_optimizer = ArenaOptimizer(a["dynamics"], lr, 1e-2, clip, layout=self._dynamics_layout())

    def _create_replay_buffer(self) -> ReplayBuffer:
        shape = (self.frame_stack, *self.obs_shape) if self.frame_stack > 1 else self.obs_shape
        return ReplayBuffer(capacity=self.training_config.buffer_size, obs_shape=shape, action_dim=self.action_dim, device=self.device, optimize_memory=True)

    def enable_data_parallel(self):
        for a in (*self.active_inference.arenas.values(), self.encoder_arena):
            parallel.broadcast_arena_(a.flat)
        for b in self.encoder.buffers():
            parallel.broadcast_arena_(b)
        self.active_inference.arenas["score"].reset_ema()
        self.active_inference.noise.seed += 7919 * parallel.rank()
        self.encoder.rng_seed += 7919 * parallel.rank()

    # -------------------------------------------------------------------------------------------------- observation plumbing
    def _process_observation(self, observation) -> torch.Tensor:
        a = observation.detach().cpu().numpy() if torch.is_tensor(observation) else np.array(observation)
        if a.ndim == 2:
            a = a[None]
        elif a.ndim == 3 and a.shape[-1] in (1, 3):
            a = a.transpose(2, 0, 1)
        elif a.ndim == 4 and a.shape[-1] in (1, 3):
            a = a.transpose(0, 3, 1, 2)
        elif a.ndim not in (3, 4):
            raise ValueError(f"Unexpected observation shape: {a.shape}")
        was_u8 = a.dtype == np.uint8
        t = torch.as_tensor(a.astype(np.float32))
        if t.dim() == 3 or (t.dim() == 4 and t.shape[0] == self.frame_stack):
            t = t.unsqueeze(0)
        if was_u8 or float(t.max()) > 1.0:
            t = t / 255.0
        return t

    def _process_batch_observations(self, observations: torch.Tensor) -> torch.Tensor:
        observations = observations.to(self.device)
        if observations.dtype == torch.uint8:
            observations = observations.float() / 255.0
        if observations.dim() == 5:
            b, fs, c, h, w = observations.shape
            observations = observations.reshape(b, fs * c, h, w)
        return observations

    def _encoder_noise(self, B: int):
        """Per-pass randomness: crop offsets of RandomShiftAugmentation (fused into the first conv's gather) + dropout masks."""
        hook = getattr(self.active_inference.noise, "encoder", None)
        n = hook(B, self.config.latent_dim, self.pixel_config.random_shift_pad) if hook is not None else None
        if n is None:
            n = {}
            if self.augmentation is not None and self.training:
                n["shift_h"], n["shift_w"] = self.augmentation.sample_shifts(B, self.device)
        if self.augmentation is None or not self.training:
            n.pop("shift_h", None); n.pop("shift_w", None)
        n["pad"] = self.pixel_config.random_shift_pad
        return n

    def encode_observation(self, observation: torch.Tensor, keep_for_backward: bool = False, slot: str = "a"):
        """pixels -> (augmentation, train mode only) -> DrQ-v2 features [B, latent_dim]   (pixel_agent.py:163-201)"""
        if observation.dim() == 3:
            observation = observation.unsqueeze(0)
        elif observation.dim() not in (4, 5):
            raise ValueError(f"Unexpected observation shape: {observation.shape}")
        observation = self._process_batch_observations(observation)
        return self.encoder(observation, self._encoder_noise(observation.shape[0]), keep_for_backward=keep_for_backward, slot=slot)

    def act(self, observation: np.ndarray, deterministic: bool = False) -> Tuple[np.ndarray, Dict[str, Any]]:
        enc = self.encode_observation(self._process_observation(observation).to(self.device))
        action_t, info = self.active_inference.act(enc, deterministic=deterministic)
        action = action_t.cpu().numpy().reshape(-1)
        if not deterministic and self.training and self.exploration_noise > 0:
            action = np.clip(action + np.random.normal(0, self.exploration_noise, size=action.shape), -1, 1)
        info["encoded_obs_norm"] = float(enc.norm())
        return action, info

    def compute_representation_loss(self, obs, next_obs, actions, latents, next_latents, d_next_obs: torch.Tensor = None, weight: float = 1.0):
        """InfoNCE between the dynamics model's predicted next latent and the encoded next observation (pixel_agent.py:413-437).
        With d_next_obs, weight * d(loss)/d(next_obs) is accumulated into it; latent_dynamics gradients are not produced (the
        reference zeroes them before they are ever used, :396)."""
        ai = self.active_inference
        if d_next_obs is None:
            d_next_obs = torch.zeros_like(next_obs)
        return nat.infonce_step(ai.latent_dynamics.table(), latents, actions.contiguous().float(), next_obs.contiguous(), weight, d_next_obs, ai._ws)[0]

    # --------------------------------------------------------------------------------------------------
    def train_step(self) -> Dict[str, float]:
        """pixel_agent.py:274-411.  encode(obs), encode(next_obs) [two spectral-norm power iterations, as in the reference] ->
        belief latents for both (one merged 2B-row reverse-diffusion loop) -> ELBO (+ feature-decoder reconstruction) and
        InfoNCE, whose feature gradients drive the two encoder backward passes -> policy / value / dynamics phases."""
        cfg, ai = self.config, self.active_inference
        B = cfg.batch_size
        ready = len(self.replay_buffer) >= B
        if parallel.is_dp():      # the early return sits in front of this step's collectives: every rank takes it or none does
            ready = bool(parallel.agree_min([ready], device=self.device)[0])
        if not ready:
            return {}
        ai.concurrent_belief_loops = self.overlap_streams       # one stream only when asked for one (profiling)
        batch = self.replay_buffer.sample(B)
        actions, rewards, dones = batch["actions"], batch["rewards"], batch["dones"]
        dp = parallel.is_dp()
        N, train = cfg.diffusion.num_diffusion_steps, ai.latent_score_network.training
        # Schedule (results unchanged; draws are made in the reference's order: encoder pass a, encoder pass b, belief a, belief b, ELBO, EFE):
        # the belief over obs needs encode(obs) only, the ELBO needs that belief only, and nothing before the value phase reads the belief over
        # next_obs.  So chain a starts on a chain stream as soon as encode(obs) is enqueued and runs underneath encode(next_obs); chain b
        # starts behind encode(next_obs) and runs underneath the ELBO, InfoNCE, the encoder backward passes and the policy phase.
        self._mark("start")
        obs_a = self._process_batch_observations(batch["observations"]); obs_b = self._process_batch_observations(batch["next_observations"])
        n_a, n_b = self._encoder_noise(B), self._encoder_noise(B)
        enc_obs, ctx_a = self.encoder(obs_a, n_a, keep_for_backward=True, slot="a")
        # (data parallel: RCCL's own stream would be a fifth hardware queue next to main + side + two chain streams - aid_amd/streams.py - so the
        # paired loop of the calling stream + one chain stream stays)
        early = self.early_belief_chains and self.overlap_streams and not dp
        nz_a = ai.noise.belief(B, ai.latent_dim, cfg.hidden_dim, N, train) if early else None
        h_a = ai.denoise_start(enc_obs, 0, nz_a) if early else None
        self._mark("main: encode(obs) done")
        enc_next, ctx_b = self.encoder(obs_b, n_b, keep_for_backward=True, slot="b")
        self._mark("main: encode(next_obs) done")
        nrew = self.reward_normalizer.update_and_normalize(rewards, parallel.allgather_cat(rewards) if dp else None)
        if h_a is not None:
            nz_b = ai.noise.belief(B, ai.latent_dim, cfg.hidden_dim, N, train)
            if self.stagger_next_obs_chain:      # chain b only once chain a is done: chain a (the critical path) runs next to encode(next_obs) alone
                latents = ai.denoise_join(h_a)
                h_b = ai.denoise_start(enc_next, 1, nz_b)
            else:
                h_b = ai.denoise_start(enc_next, 1, nz_b)
                latents = ai.denoise_join(h_a)
            next_latents = None if h_b is not None else ai.denoise_only(enc_next, noise=nz_b, ws=ai._ws)
        else:
            h_b = None
            both = ai.denoise_pair(enc_obs, enc_next, nz_a, ai.noise.belief(B, ai.latent_dim, cfg.hidden_dim, N, train)) if nz_a is not None else ai.denoise_pair(enc_obs, enc_next)
            latents, next_latents = both[:B], both[B:]
        self._mark("main: belief over obs joined")
        # score / diffusion / encoder / feature-decoder group.  Schedule (results unchanged up to which GEMM variant a stream's dispatch hint selects):
        # once the belief over obs is known, two chains of small dependent kernels are ready at the same time - the ELBO (forward, backward, gradient-penalty
        # double backward: ~400 launches) and the EFE rollout + BPTT - and they overlap well with EACH OTHER (both leave most CUs idle), whereas either of
        # them crawls underneath the encoder backward's chip-filling convolutions (a 7-us kernel waits for workgroup slots behind 100-us conv workgroups).
        # So the ELBO goes to the side stream next to the EFE on the calling stream, and the two encoder backward passes + the optimiser step follow the ELBO on
        # the side stream.  The host draws in the reference's order (ELBO, then EFE).  InfoNCE (no draws) runs first: it needs only features and latents.
        self.score_optimizer.zero_grad()
        main = torch.cuda.current_stream()
        side = self._comm_stream() if self.overlap_streams else main
        d_next = torch.zeros_like(enc_next)
        side_elbo = self.elbo_on_side_stream and self.overlap_streams
        if side_elbo:
            contrastive = self.compute_representation_loss(enc_obs, enc_next, actions, latents, next_latents, d_next, cfg.contrastive_weight)
            self._mark("main: InfoNCE done")
        else:
            elbo_loss, elbo_info = ai.compute_diffusion_elbo(enc_obs, nrew, latents)
            contrastive = self.compute_representation_loss(enc_obs, enc_next, actions, latents, next_latents, d_next, cfg.contrastive_weight)
            self._mark("main: ELBO + InfoNCE done")
        self._hint_main_stream(main, 2 if (self.overlap_streams and self.main_shares_cus) else 0)      # aid_stream_hint bit 1, as in the state agent
        if self.overlap_streams:
            side.wait_stream(main)
            for t_ in (enc_obs, enc_next, nrew, latents, d_next):
                t_.record_stream(side)
        with torch.cuda.stream(side):
            if side_elbo:
                elbo_loss, elbo_info = ai.compute_diffusion_elbo(enc_obs, nrew, latents, ws=self._side_workspace())
                self._mark("side: ELBO done", side)
            if dp and self.dp_overlap_encoder_allreduce:
                # data parallel: the head phase of BOTH passes first (they accumulate into the same Linear(451 584 -> 2L) gradient), then that gradient's
                # all-reduce is started and the attention / conv-stack phases run underneath it (each pass keeps its hand-over buffer in its own workspace)
                ws_b = self._encoder_ws_b()
                self.encoder.backward(ai._last_d_obs, ctx_a, phase=1)
                self.encoder.backward(d_next, ctx_b, phase=1, ws=ws_b)
                self.score_optimizer.allreduce_encoder_head()
                self.encoder.backward(ai._last_d_obs, ctx_a, phase=2)
                self.encoder.backward(d_next, ctx_b, phase=2, ws=ws_b)
            else:
                self.encoder.backward(ai._last_d_obs, ctx_a)
                self.encoder.backward(d_next, ctx_b)
            self._mark("side: encoder backward x2 done", side)
            # chain b (the belief over next_obs) reads the score-network weights in place on its chain stream: the AdamW / EMA step must not
            # rewrite them before that loop has finished (the reference computes next_latents before score_optimizer.step(), pixel_agent.py:307 vs
            # :332).  Only the step waits; the encoder backward passes above still overlap the chain.
            if h_b is not None:
                side.wait_stream(h_b["stream"])
            self.score_optimizer.step()
            self._mark("side: score / encoder step done", side)
        if self.overlap_streams:
            ai._last_d_obs.record_stream(side)
            for v in elbo_info.values():
                if torch.is_tensor(v):
                    v.record_stream(main)
        # policy
        self.policy_optimizer.zero_grad()
        efe, efe_info = ai.compute_expected_free_energy_diffusion(latents, horizon=cfg.efe_horizon, backward=True)
        self.policy_optimizer.step()
        self._mark("main: EFE + policy step done")
        # value (the pixel agent reuses next_latents, pixel_agent.py:364-371) and dynamics
        if h_b is not None:
            next_latents = ai.denoise_join(h_b)
        self._mark("main: belief over next_obs joined")
        self.value_optimizer.zero_grad()
        nl_x, nrew_x, dones_x = parallel.halo_from_next_rank(next_latents, nrew, dones, rows=5) if dp else (next_latents, nrew, dones)
        value_loss, _ = nat.value_step(ai.value_network.table(), ai.value_network.grad_table(), latents, nl_x, nrew_x, dones_x,
                                       cfg.discount_factor, 0.95, 5, ws=ai._ws)
        self.value_optimizer.step()
        self.dynamics_optimizer.zero_grad()
        dyn_loss = nat.dynamics_step(ai.latent_dynamics.table(), ai.latent_dynamics.grad_table(), latents, actions, next_latents, ws=ai._ws)
        self.dynamics_optimizer.step()
        self._mark("main: value + dynamics done")
        if self.overlap_streams:
            main.wait_stream(side)        # the next encoder forward needs the updated encoder
        self.total_steps += 1
        keys, vals = [], []
        for k, v in {**elbo_info, "contrastive_loss": contrastive, "policy_loss": efe_info.pop("policy_loss"), **efe_info, "value_loss": value_loss,
                     "dynamics_loss": dyn_loss}.items():
            if torch.is_tensor(v):
                keys.append(k); vals.append(v.reshape(1))
        metrics = dict(zip(keys, torch.cat(vals).cpu().tolist()))
        metrics.update({"num_trajectories": efe_info["num_trajectories"], "horizon": efe_info["horizon"]})
        self._last = {"latents": latents, "next_latents": next_latents, "encoded_obs": enc_obs, "encoded_next_obs": enc_next}
        return metrics

    overlap_streams = True    # False: every phase on the current stream, in order
    timeline = None           # perf diagnostics (tools/perf/timeline_cfg4.py): a list to which train_step appends (label, event, host time)

    def _mark(self, label, stream=None):
        if self.timeline is not None:
            import time as _t
            ev = torch.cuda.Event(enable_timing=True); ev.record(stream or torch.cuda.current_stream()); self.timeline.append((label, ev, _t.perf_counter()))
    stagger_next_obs_chain = False      # chain b only after chain a (measured slower: 63.8 vs 60.5 ms, tools/perf/README.md round 5)
    elbo_on_side_stream = True          # the ELBO on the side stream next to the EFE on the calling stream (see train_step); False: ELBO on the calling stream, EFE behind it

    dp_overlap_encoder_allreduce = True   # data parallel: all-reduce the encoder head's gradient underneath the conv stack's backward (see train_step)

    def _encoder_ws_b(self):
        if getattr(self, "_enc_ws_b", None) is None:
            self._enc_ws_b = nat.Workspace(self.device)
        return self._enc_ws_b

    def _side_workspace(self):
        if getattr(self, "_ws_side", None) is None:
            self._ws_side = nat.Workspace(self.device)
        return self._ws_side
    early_belief_chains = True   # the two belief updates as separately started chains (see train_step); False: one merged / paired loop after both encoder passes

    def _comm_stream(self):
        if getattr(self, "_comm", None) is None:
            # the process-wide side stream (aid_amd/streams.py: a process has four hardware queues, so every agent shares the same three extra streams)
            from ..streams import side_stream
            self._comm = side_stream(self.device)
        return self._comm


PixelBasedAgent = DiffusionPixelAgent
