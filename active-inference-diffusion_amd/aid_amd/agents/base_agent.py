"""BaseActiveInferenceAgent and the reward normaliser
(active_inference_diffusion/agents/base_agent.py:24-185)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Tuple

import numpy as np
import torch

from .. import _native as nat
from ..arena import AdamL2Optimizer, ArenaOptimizer
from ..configs.config import ActiveInferenceConfig, TrainingConfig
from ..core.active_inference import EMAModel


class RunningMeanStd:
    """Welford-merge running statistics (base_agent.py:24-52) kept in fp64 ON THE DEVICE: update + normalise is one
    small kernel, removing the two device<->host round trips of the reference's train_step (state_agent.py:126-133)."""

    def __init__(self, epsilon: float = 1e-4, shape=(), device="cuda"):
        assert shape == (), "only scalar statistics are used by the reference"
        self.state = torch.tensor([0.0, 1.0, float(epsilon)], dtype=torch.float64, device=device)

    mean = property(lambda self: float(self.state[0]))
    var = property(lambda self: float(self.state[1]))
    count = property(lambda self: float(self.state[2]))

    def update_and_normalize(self, rewards: torch.Tensor, global_rewards: torch.Tensor = None) -> torch.Tensor:
        """update(x) followed by normalize(x), as train_step does; float32 out.  Data parallel: pass the all-gathered
        rewards of every rank so the statistics are those of the global batch."""
        if global_rewards is not None:
            return nat.reward_normalize_global(global_rewards.contiguous().float(), rewards.contiguous().float(), self.state)
        return nat.reward_normalize(rewards.contiguous().float(), self.state)

    def update(self, x):
        self.update_and_normalize(torch.as_tensor(np.asarray(x, dtype=np.float32)).to(self.state.device))

    def normalize(self, x):
        x = np.asarray(x)
        return (x - self.mean) / np.sqrt(self.var + 1e-8)


class BaseActiveInferenceAgent(ABC):
    def __init__(self, env, config: ActiveInferenceConfig, training_config: TrainingConfig):
        self.env, self.config, self.training_config = env, config, training_config
        self.device = torch.device(config.device)
        self._setup_dimensions()
        self._build_models()
        ai = self.active_inference
        self.score_ema = EMAModel(ai.latent_score_network, decay=0.9999, device=self.device, arena=ai.arenas["score"], prefix="latent_score_network.")
        self._setup_optimizers()
        self.replay_buffer = self._create_replay_buffer()
        self.total_steps, self.episode_count = 0, 0
        self.exploration_noise = training_config.exploration_noise
        self.reward_normalizer = RunningMeanStd(shape=(), device=self.device)

    @abstractmethod
    def _setup_dimensions(self): ...

    @abstractmethod
    def _build_models(self): ...

    @abstractmethod
    def _create_replay_buffer(self): ...

    @abstractmethod
    def _process_observation(self, observation: np.ndarray) -> torch.Tensor: ...

    @abstractmethod
    def _process_batch_observations(self, observations: torch.Tensor) -> torch.Tensor: ...

    main_shares_cus = True    # with the side stream in use, the main stream's 512-wide full-row kernels take 16-deep stages (74 KB of LDS) so that the
                              # reverse loop's 80 KB workgroups can share their CUs: -0.7 ms per cfg-2 step (alone those kernels are 0.2 % slower)

    def _hint_main_stream(self, main, flags: int):
        key = (main.cuda_stream, flags)
        prev = getattr(self, "_main_hint", None)
        if prev != key:
            if prev is not None and prev[0] != key[0] and prev[1]:
                nat.stream_hint(prev[0], 0)          # the stream used before (raw handle): its slot in the library's table is freed
            nat.stream_hint(main, flags)
            self._main_hint = key

    def _setup_optimizers(self):
        """Four AdamW groups (state_agent.py:245-274): lr = config.learning_rate; weight decay 1e-5 on the score and policy
        groups, torch's default 1e-2 on value and dynamics (the reference passes none); clip-by-global-norm = gradient_clip;
        score-net EMA 0.9999 fused into the score step."""
        ai = self.active_inference
        a, lr, clip = ai.arenas, self.config.learning_rate, self.config.gradient_clip
        self.score_optimizer = ArenaOptimizer(a["score"], lr, 1e-5, clip, ema_decay=0.9999, layout=self._score_layout())
        self.policy_optimizer = ArenaOptimizer(a["policy"], lr, 1e-5, clip)
        self.value_optimizer = ArenaOptimizer(a["value"], lr, 1e-2, clip)
        self.dynamics_optimizer = ArenaOptimizer(a["dynamics"], lr, 1e-2, clip, layout=self._dynamics_layout())
        if "epistemic" in a:   # base_agent.py:134-139 (the reference's concrete agents forget to create it: SURVEY D3)
            self.epistemic_optimizer = AdamL2Optimizer(a["epistemic"], lr * 0.1, 1e-5, clip)
            self.active_inference.epistemic_optimizer = self.epistemic_optimizer

    # Checkpoint layouts: the parameter lists the reference passes to torch.optim.AdamW, in its order (state_agent.py:245-274).  The score
    # arena stores its tensors in another order (the 13 adaLN matrices back to back); the dynamics optimiser also lists the observation
    # decoder and the reward predictor, which it never updates - positions without state.
    def _score_layout(self):
        ai, sc = self.active_inference, self.active_inference.arenas["score"]
        return ([(sc, "latent_score_network." + n) for n, _ in ai.latent_score_network.named_parameters()] +
                [(sc, "latent_diffusion." + n) for n, _ in ai.latent_diffusion.named_parameters()])

    def _dynamics_layout(self):
        ai, dy = self.active_inference, self.active_inference.arenas["dynamics"]
        n_frozen = sum(1 for _ in ai.observation_decoder.parameters()) + sum(1 for _ in ai.reward_predictor.parameters())
        return [(dy, "latent_dynamics." + n) for n, _ in ai.latent_dynamics.named_parameters()] + [None] * n_frozen

    def act(self, observation: np.ndarray, deterministic: bool = False) -> Tuple[np.ndarray, Dict[str, Any]]:
        obs = self._process_observation(observation).to(self.device)
        action_t, info = self.active_inference.act(obs, deterministic=deterministic)
        action = action_t.cpu().numpy().squeeze()
        if not deterministic and self.training and self.exploration_noise > 0:
            action = np.clip(action + np.random.normal(0, self.exploration_noise, size=action.shape), -1, 1)
        return action, info

    def train_step(self) -> Dict[str, float]:
        raise NotImplementedError

    def update_exploration(self):
        self.exploration_noise = max(self.exploration_noise * self.training_config.exploration_decay, self.training_config.min_exploration)

    @property
    def training(self) -> bool:
        return self.active_inference.training
