"""Test helper: a NoiseSource that replays the reference's draws (oracle.noise.NoiseStream, reference call order)
into the layouts the native entry points take.  Lives under tests/ — the product never imports the oracle."""
import numpy as np
import torch

from aid_amd.noise import NoiseSource
from oracle import aid_oracle as orc
from oracle.noise import NoiseStream


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda().contiguous()


class ParityNoise(NoiseSource):
    def __init__(self, key: int, pixel: bool = False, burn_epistemic_samples: int = 0):
        super().__init__(seed=0)
        self.ns = NoiseStream(key)
        self.burn_epistemic_samples = burn_epistemic_samples   # >0: consume the reference's pixel-estimator draws (alignment with reference goldens)
        self.live_epistemic_samples = 0                        # >0: keep them (the pixel estimator is evaluated: ai.pixel_epistemic = True)
        self.pixel = pixel        # pixel mode: feature-decoder mask instead of the 3 decoder masks, no estimator draws in the EFE

    def encoder(self, B, L, pad):
        """One DrQ-v2 pass in train mode: RandomShiftAugmentation offsets, Dropout2d masks of layers 1, 2, Dropout mask of the head."""
        ns = self.ns
        sh, sw = ns.randint(0, 2 * pad + 1, B), ns.randint(0, 2 * pad + 1, B)
        i32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int32)).cuda()
        return {"shift_h": i32(sh), "shift_w": i32(sw), "drop2d_1": dev(ns.dropout_mask((B, 64), 0.025)), "drop2d_2": dev(ns.dropout_mask((B, 128), 0.05)),
                "drop_out": dev(ns.dropout_mask((B, 2 * L), 0.1))}

    def rekey(self, key: int):
        self.ns = NoiseStream(key)

    def replay_indices(self, size, batch):
        return self.ns.randint(0, size, batch)

    def belief(self, B, L, D, N, train):
        ns = self.ns
        z0 = ns.randn(B, L)
        eps = np.zeros((N, B, L), np.float32)
        masks = np.ones((N, B, D), np.float32) if train else None
        for t in reversed(range(N)):
            if train:
                masks[t] = ns.dropout_mask((B, D), 0.1)
            if t > 0:
                eps[t] = ns.randn(B, L)
        fd = None
        if self.pixel:
            dec = None
            if train:
                fd = dev(ns.dropout_mask((B, D), 0.2))  # feature-decoder mask of the reconstruction diagnostic (active_inference.py:292-295)
        else:
            dec = [dev(ns.dropout_mask((B, w), 0.2)) for w in (2 * D, 2 * D, D)] if train else None   # decode_observation masks
        return {"z_init": dev(z0), "step_noise": dev(eps), "drop_masks": dev(masks) if train else None, "dec_masks": dec, "fd_mask": fd}

    def elbo(self, B, L, D, first_call, time_importance):
        ns = self.ns
        if self.pixel:
            n = {"fd_mask": ns.dropout_mask((B, D), 0.2)}
        else:
            n = {"dec_mask0": ns.dropout_mask((B, 2 * D), 0.2), "dec_mask1": ns.dropout_mask((B, 2 * D), 0.2), "dec_mask2": ns.dropout_mask((B, D), 0.2)}
        n["t"] = ns.rand(B) if first_call else orc.importance_sample_time(time_importance.detach().cpu(), B, ns).numpy()
        n["eps"] = ns.randn(B, L); n["score_mask"] = ns.dropout_mask((B, D), 0.1); n["gp_mask"] = ns.dropout_mask((B, D), 0.1)
        n["prior_eps"] = ns.randn(B, L)
        return {k: dev(v) for k, v in n.items()}

    def pixel_estimator(self, B, L, S, train=True):
        """One pixel-mode estimator forward (K = 1): the draws of burn_pixel_epistemic_draws, kept."""
        ns = self.ns
        R = S * B
        n = {"mine_z_eps": np.concatenate([ns.randn(B, L) for _ in range(S)]), "mine_delta": np.zeros((4, R, L), np.float32)}
        am = np.ones((4, R * 8, 16, 441), np.float32)
        for j in range(4):
            n["mine_delta"][j] = ns.randn(R, L)
            if train:
                am[j] = ns.dropout_mask((R * 8, 16, 441), 0.1)
        n["attn_mask"] = am
        n["proj_mask"] = ns.dropout_mask((R, 512), 0.1) if train else np.ones((R, 512), np.float32)
        mm = [ns.dropout_mask((R, 512), 0.1) if train else np.ones((R, 512), np.float32) for _ in range(2)]
        perms = np.stack([ns.randperm(B) for _ in range(S)]).astype(np.int32)
        mm += [ns.dropout_mask((R, 512), 0.1) if train else np.ones((R, 512), np.float32) for _ in range(2)]
        n["mine_mask"] = np.stack(mm)
        out = {k: dev(v) for k, v in n.items()}
        out["perms"] = torch.from_numpy(perms).cuda().contiguous()
        return out

    def mine_train(self, B, L, S):
        """train_epistemic_estimator: the draws of ONE estimator forward (same order as inside the EFE rollout, K = 1)."""
        ns = self.ns
        R = S * B
        n = {"mine_z_eps": np.concatenate([ns.randn(B, L) for _ in range(S)]), "mine_delta": np.stack([ns.randn(R, L) for _ in range(4)])}
        n["proj_mask"] = ns.dropout_mask((R, 512), 0.1)
        mm = [ns.dropout_mask((R, 512), 0.1), ns.dropout_mask((R, 512), 0.1)]
        perms = np.stack([ns.randperm(B) for _ in range(S)]).astype(np.int32)
        mm += [ns.dropout_mask((R, 512), 0.1), ns.dropout_mask((R, 512), 0.1)]
        n["mine_mask"] = np.stack(mm)
        out = {k: dev(v) for k, v in n.items()}
        out["perms"] = torch.from_numpy(perms).cuda().contiguous()
        return out

    def efe(self, B, L, A, H, K, S):
        ns = self.ns
        if self.pixel:
            n = {"policy_eps": np.zeros((H, K * B, A), np.float32), "reparam_eps": np.zeros((H, K * B, L), np.float32)}
            live = self.live_epistemic_samples
            R1 = live * B
            pe = [{"mine_z_eps": np.zeros((K * R1, L), np.float32), "mine_delta": np.zeros((4, K * R1, L), np.float32),
                   "attn_mask": np.ones((4, K * R1 * 8, 16, 441), np.float32), "proj_mask": np.ones((K * R1, 512), np.float32),
                   "mine_mask": np.ones((4, K * R1, 512), np.float32), "perms": np.zeros((K * live, B), np.int32)} for _ in range(H)] if live else None
            for k in range(K):
                for t in range(H):
                    n["policy_eps"][t, k * B:(k + 1) * B] = ns.randn(B, A)
                    n["reparam_eps"][t, k * B:(k + 1) * B] = ns.randn(B, L)
                    if live:      # the pixel estimator's draws for this (k, t), reference order, scattered into trajectory k's rows
                        d, rs = pe[t], slice(k * R1, (k + 1) * R1)
                        for s_ in range(live):
                            d["mine_z_eps"][k * R1 + s_ * B:k * R1 + (s_ + 1) * B] = ns.randn(B, L)
                        for j in range(4):
                            d["mine_delta"][j, rs] = ns.randn(R1, L)
                            d["attn_mask"][j, k * R1 * 8:(k + 1) * R1 * 8] = ns.dropout_mask((R1 * 8, 16, 441), 0.1)
                        d["proj_mask"][rs] = ns.dropout_mask((R1, 512), 0.1)
                        d["mine_mask"][0, rs] = ns.dropout_mask((R1, 512), 0.1); d["mine_mask"][1, rs] = ns.dropout_mask((R1, 512), 0.1)
                        for s_ in range(live):
                            d["perms"][k * live + s_] = ns.randperm(B)
                        d["mine_mask"][2, rs] = ns.dropout_mask((R1, 512), 0.1); d["mine_mask"][3, rs] = ns.dropout_mask((R1, 512), 0.1)
                    elif self.burn_epistemic_samples:
                        orc.burn_pixel_epistemic_draws(ns, B, L, self.burn_epistemic_samples, True)
            out = {k: dev(v) for k, v in n.items()}
            if live:
                out["pixel_epistemic"] = [{k: (torch.from_numpy(v).cuda().contiguous() if k == "perms" else dev(v)) for k, v in d.items()} for d in pe]
            return out
        M, R = K * B, K * S * B
        n = {"policy_eps": np.zeros((H, M, A), np.float32), "reparam_eps": np.zeros((H, M, L), np.float32),
             "mine_z_eps": np.zeros((H, R, L), np.float32), "mine_delta": np.zeros((4, H, R, L), np.float32),
             "proj_mask": np.zeros((H, R, 512), np.float32), "mine_mask": np.zeros((4, H, R, 512), np.float32),
             "perms": np.zeros((H, K, S, B), np.int32)}
        for k in range(K):
            rs = slice(k * S * B, (k + 1) * S * B)
            for t in range(H):
                n["policy_eps"][t, k * B:(k + 1) * B] = ns.randn(B, A)
                n["reparam_eps"][t, k * B:(k + 1) * B] = ns.randn(B, L)
                for s_ in range(S):
                    n["mine_z_eps"][t, k * S * B + s_ * B:k * S * B + (s_ + 1) * B] = ns.randn(B, L)
                for j in range(4):
                    n["mine_delta"][j, t, rs] = ns.randn(S * B, L)
                n["proj_mask"][t, rs] = ns.dropout_mask((S * B, 512), 0.1)
                n["mine_mask"][0, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
                n["mine_mask"][1, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
                for s_ in range(S):
                    n["perms"][t, k, s_] = ns.randperm(B)
                n["mine_mask"][2, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
                n["mine_mask"][3, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
        return {k: (torch.from_numpy(v).cuda().contiguous() if k == "perms" else dev(v)) for k, v in n.items()}
