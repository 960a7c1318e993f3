"""ctypes binding of libaid_hip.so (the C ABI declared in include/aid_hip.h).

PyTorch is used for device memory, streams and torch.distributed only; every arithmetic op of the
hot path goes through this module into hand-written gfx950 kernels.  There is NO CPU fallback:
if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaid_hip.so")
AID_MAX_BLOCKS = 8
T_DISCRETE, T_CONTINUOUS = 0, 1
ACT = {"none": 0, "relu": 1, "silu": 2, "gelu": 3, "mish": 4, "tanh": 5}
PROF_GEMM, PROF_ROWWISE, PROF_OPTIM, PROF_GEMM_TILE = 0, 1, 2, 3

_fp = C.POINTER(C.c_float)


class AidDiTBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ada1_w", "ada1_b", "ada2_w", "ada2_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
        "mlp0_w", "mlp0_b", "mlp2_w", "mlp2_b")]


_SCORE_PTRS = ("time_scale", "output_multiplier", "te_freq_scale", "te1_w", "te1_b", "te3_w", "te3_b",
               "oe0_w", "oe0_b", "oe1_g", "oe1_b", "oe4_w", "oe4_b", "oe5_g", "oe5_b", "oe7_w", "oe7_b", "oe8_g", "oe8_b",
               "cte0_w", "cte0_b", "cte2_w", "cte2_b", "cte4_w", "cte4_b", "lp_w", "lp_b")


class AidScoreNet(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("L", "O", "D", "TE", "n_blocks", "_pad")]
                + [(n, C.c_void_p) for n in _SCORE_PTRS]
                + [("blk", AidDiTBlock * AID_MAX_BLOCKS)]
                + [(n, C.c_void_p) for n in ("adaf_w", "adaf_b", "op0_w", "op0_b", "op2_w")])


# state_dict key (relative to the score-net prefix) -> struct field
SCORE_KEYMAP = {
    "time_scale": "time_scale", "output_multiplier": "output_multiplier", "time_embed.0.freq_scale": "te_freq_scale",
    "time_embed.1.weight": "te1_w", "time_embed.1.bias": "te1_b", "time_embed.3.weight": "te3_w", "time_embed.3.bias": "te3_b",
    "obs_encoder.0.weight": "oe0_w", "obs_encoder.0.bias": "oe0_b", "obs_encoder.1.weight": "oe1_g", "obs_encoder.1.bias": "oe1_b",
    "obs_encoder.4.weight": "oe4_w", "obs_encoder.4.bias": "oe4_b", "obs_encoder.5.weight": "oe5_g", "obs_encoder.5.bias": "oe5_b",
    "obs_encoder.7.weight": "oe7_w", "obs_encoder.7.bias": "oe7_b", "obs_encoder.8.weight": "oe8_g", "obs_encoder.8.bias": "oe8_b",
    "continuous_time_embed.0.weight": "cte0_w", "continuous_time_embed.0.bias": "cte0_b",
    "continuous_time_embed.2.weight": "cte2_w", "continuous_time_embed.2.bias": "cte2_b",
    "continuous_time_embed.4.weight": "cte4_w", "continuous_time_embed.4.bias": "cte4_b",
    "latent_proj.weight": "lp_w", "latent_proj.bias": "lp_b",
    "norm_final.adaLN_modulation.1.weight": "adaf_w", "norm_final.adaLN_modulation.1.bias": "adaf_b",
    "output_proj.0.weight": "op0_w", "output_proj.0.bias": "op0_b", "output_proj.2.weight": "op2_w",
}
BLOCK_KEYMAP = {
    "norm1.adaLN_modulation.1.weight": "ada1_w", "norm1.adaLN_modulation.1.bias": "ada1_b",
    "norm2.adaLN_modulation.1.weight": "ada2_w", "norm2.adaLN_modulation.1.bias": "ada2_b",
    "attention.in_proj_weight": "in_proj_w", "attention.in_proj_bias": "in_proj_b",
    "attention.out_proj.weight": "out_proj_w", "attention.out_proj.bias": "out_proj_b",
    "mlp.0.weight": "mlp0_w", "mlp.0.bias": "mlp0_b", "mlp.2.weight": "mlp2_w", "mlp.2.bias": "mlp2_b",
}


class AidLinear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("in_", C.c_int32), ("out", C.c_int32)]


class AidLN(C.Structure):
    _fields_ = [("g", C.c_void_p), ("b", C.c_void_p)]


class AidPolicyNet(C.Structure):
    _fields_ = [("L", C.c_int32), ("A", C.c_int32), ("D", C.c_int32), ("_pad", C.c_int32),
                ("enc0", AidLinear), ("enc_ln", AidLN), ("enc3", AidLinear), ("trunk", AidLinear * 3), ("trunk_ln", AidLN * 3),
                ("mean0", AidLinear), ("mean2", AidLinear), ("lstd0", AidLinear), ("lstd2", AidLinear)]


class AidValueNet(C.Structure):
    _fields_ = [("L", C.c_int32), ("D", C.c_int32), ("TE", C.c_int32), ("_pad", C.c_int32), ("freq_scale", C.c_void_p),
                ("te1", AidLinear), ("net", AidLinear * 3), ("ln", AidLN * 3), ("out", AidLinear)]


class AidDynamicsNet(C.Structure):
    _fields_ = [("L", C.c_int32), ("A", C.c_int32), ("D", C.c_int32), ("_pad", C.c_int32),
                ("net", AidLinear * 3), ("ln", AidLN * 3), ("out", AidLinear)]


class AidRewardNet(C.Structure):
    _fields_ = [("L", C.c_int32), ("D", C.c_int32), ("l0", AidLinear), ("ln", AidLN), ("l3", AidLinear), ("l5", AidLinear)]


class AidDecoderNet(C.Structure):
    _fields_ = [("L", C.c_int32), ("O", C.c_int32), ("D", C.c_int32), ("_pad", C.c_int32),
                ("s", AidLinear * 3), ("ln", AidLN * 3), ("out", AidLinear)]


class AidEpistemicNet(C.Structure):
    _fields_ = [("L", C.c_int32), ("O", C.c_int32), ("perturbation_scale", C.c_void_p),
                ("fe0", AidLinear), ("fe2", AidLinear), ("fe4", AidLinear), ("jp0", AidLinear), ("jp_ln", AidLN), ("jp4", AidLinear),
                ("lp0", AidLinear), ("lp2", AidLinear), ("mn0", AidLinear), ("mn3", AidLinear), ("mn6", AidLinear)]


class AidEncoderNet(C.Structure):
    _fields_ = [("feature_dim", C.c_int32), ("in_ch", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("conv_w", C.c_void_p * 4), ("conv_u", C.c_void_p * 4), ("conv_v", C.c_void_p * 4), ("gn_g", C.c_void_p * 4), ("gn_b", C.c_void_p * 4),
                ("att_temperature", C.c_void_p), ("att_conv_w", C.c_void_p), ("att_conv_b", C.c_void_p), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
                ("out0", AidLinear), ("out1", AidLN), ("out4", AidLinear), ("out5", AidLN)]


class AidDiffusionParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("prior_mean", "prior_log_std", "log_snr_min", "log_snr_max")]


class AidElboNoise(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t", "eps", "prior_eps", "score_mask", "gp_mask", "dec_mask0", "dec_mask1", "dec_mask2", "d_obs_out", "mean_t")]


class AidSNConv(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w", "u", "v", "bias")]


class AidDecoderBlock(C.Structure):
    _fields_ = [("main0", AidSNConv), ("main1", AidSNConv), ("res", AidSNConv)]


class AidFeatureDecoder(C.Structure):
    _fields_ = [("l0", AidLinear), ("ln", AidLN), ("l4", AidLinear)]


class AidEfeNoise(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("policy_eps", "reparam_eps", "mine_z_eps", "mine_delta", "proj_mask", "mine_mask", "perms")]


class AidEfeCfg(C.Structure):
    _fields_ = [("H", C.c_int32), ("K", C.c_int32), ("S", C.c_int32), ("train", C.c_int32), ("skip_epistemic", C.c_int32), ("_pad", C.c_int32),
                ("epistemic_weight", C.c_float), ("pragmatic_weight", C.c_float), ("consistency_weight", C.c_float), ("discount", C.c_float),
                ("temperature", C.c_float), ("mine_alpha", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64)]


class AidElboCfg(C.Structure):
    _fields_ = [("kl_weight", C.c_float), ("diffusion_weight", C.c_float), ("reward_weight", C.c_float), ("gp_weight", C.c_float),
                ("train", C.c_int32), ("first_call", C.c_int32), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("defer_time_importance", C.c_int32), ("_pad", C.c_int32)]


def _pins_params(fn):
    """The parameter tables hold raw device pointers; keep the tensors they were built from alive for as long as the table object lives (a table built from a
    temporary dict would otherwise point at freed - and soon reused - memory)."""
    import functools

    @functools.wraps(fn)
    def wrapped(params, *a, **k):
        t = fn(params, *a, **k)
        t._keep = [v for v in params.values() if isinstance(v, torch.Tensor)]
        return t
    return wrapped


def _lin(params, name) -> AidLinear:
    w = params[name + ".weight"]; b = params.get(name + ".bias")
    l = AidLinear(); l.w = _ptr(w).value; l.b = _ptr(b).value if b is not None else None
    l.in_, l.out = w.shape[1], w.shape[0]
    return l


def _ln(params, name) -> AidLN:
    n = AidLN(); n.g = _ptr(params[name + ".weight"]).value; n.b = _ptr(params[name + ".bias"]).value
    return n


@_pins_params
def policy_table(params, prefix="policy_network.") -> AidPolicyNet:
    t = AidPolicyNet()
    t.L = params[prefix + "latent_encoder.0.weight"].shape[1]; t.D = params[prefix + "latent_encoder.0.weight"].shape[0]
    t.A = params[prefix + "mean_head.2.weight"].shape[0]
    t.enc0 = _lin(params, prefix + "latent_encoder.0"); t.enc_ln = _ln(params, prefix + "latent_encoder.1"); t.enc3 = _lin(params, prefix + "latent_encoder.3")
    for i, k in enumerate((0, 3, 6)):
        t.trunk[i] = _lin(params, prefix + f"trunk.{k}"); t.trunk_ln[i] = _ln(params, prefix + f"trunk.{k + 1}")
    t.mean0 = _lin(params, prefix + "mean_head.0"); t.mean2 = _lin(params, prefix + "mean_head.2")
    t.lstd0 = _lin(params, prefix + "log_std_head.0"); t.lstd2 = _lin(params, prefix + "log_std_head.2")
    return t


@_pins_params
def value_table(params, prefix="value_network.") -> AidValueNet:
    t = AidValueNet()
    t.TE = params[prefix + "time_embed.1.weight"].shape[0]; t.D = params[prefix + "network.0.weight"].shape[0]
    t.L = params[prefix + "network.0.weight"].shape[1] - t.TE
    t.freq_scale = _ptr(params[prefix + "time_embed.0.freq_scale"]).value
    t.te1 = _lin(params, prefix + "time_embed.1")
    for i, k in enumerate((0, 3, 6)):
        t.net[i] = _lin(params, prefix + f"network.{k}"); t.ln[i] = _ln(params, prefix + f"network.{k + 1}")
    t.out = _lin(params, prefix + "network.9")
    return t


@_pins_params
def dynamics_table(params, A: int, prefix="latent_dynamics.") -> AidDynamicsNet:
    t = AidDynamicsNet()
    t.D = params[prefix + "network.0.weight"].shape[0]; t.A = A; t.L = params[prefix + "network.0.weight"].shape[1] - A
    for i, k in enumerate((0, 3, 6)):
        t.net[i] = _lin(params, prefix + f"network.{k}"); t.ln[i] = _ln(params, prefix + f"network.{k + 1}")
    t.out = _lin(params, prefix + "network.9")
    return t


@_pins_params
def reward_table(params, prefix="reward_predictor.") -> AidRewardNet:
    t = AidRewardNet()
    t.L = params[prefix + "0.weight"].shape[1]; t.D = params[prefix + "0.weight"].shape[0]
    t.l0 = _lin(params, prefix + "0"); t.ln = _ln(params, prefix + "1"); t.l3 = _lin(params, prefix + "3"); t.l5 = _lin(params, prefix + "5")
    return t


@_pins_params
def decoder_table(params, prefix="observation_decoder.") -> AidDecoderNet:
    t = AidDecoderNet()
    t.L = params[prefix + "0.0.weight"].shape[1]; t.O = params[prefix + "3.weight"].shape[0]; t.D = params[prefix + "3.weight"].shape[1]
    for i in range(3):
        t.s[i] = _lin(params, prefix + f"{i}.0"); t.ln[i] = _ln(params, prefix + f"{i}.1")
    t.out = _lin(params, prefix + "3")
    return t


@_pins_params
def epistemic_table(params, prefix="epistemic_estimator.") -> AidEpistemicNet:
    t = AidEpistemicNet()
    t.L = params[prefix + "latent_processor.0.weight"].shape[1]; t.O = params[prefix + "feature_extractor.0.weight"].shape[1]
    t.perturbation_scale = _ptr(params[prefix + "perturbation_scale"]).value
    t.fe0, t.fe2, t.fe4 = (_lin(params, prefix + f"feature_extractor.{k}") for k in (0, 2, 4))
    t.jp0 = _lin(params, prefix + "jacobian_projector.0"); t.jp_ln = _ln(params, prefix + "jacobian_projector.1"); t.jp4 = _lin(params, prefix + "jacobian_projector.4")
    t.lp0, t.lp2 = _lin(params, prefix + "latent_processor.0"), _lin(params, prefix + "latent_processor.2")
    t.mn0, t.mn3, t.mn6 = (_lin(params, prefix + f"mine_network.{k}") for k in (0, 3, 6))
    return t


@_pins_params
def diffusion_table(params, prefix="latent_diffusion.") -> AidDiffusionParams:
    t = AidDiffusionParams()
    t.prior_mean = _ptr(params[prefix + "latent_prior_mean"]).value; t.prior_log_std = _ptr(params[prefix + "latent_prior_log_std"]).value
    t.log_snr_min = _ptr(params[prefix + "log_snr_min"]).value; t.log_snr_max = _ptr(params[prefix + "log_snr_max"]).value
    return t

_lib = None


def lib() -> C.CDLL:
    """Load libaid_hip.so once; fail loudly (no fallback) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("AID_HIP_LIB") or LIB_PATH     # another build of the same library (A/B measurements of kernel variants)
    if not os.path.exists(path):
        raise RuntimeError(
            f"aid_amd: native library not found at {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C active-inference-diffusion_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(path)
    L.aid_last_error.restype = C.c_char_p
    L.aid_version.restype = C.c_int
    for name in ("aid_score_forward_workspace", "aid_denoise_loop_workspace", "aid_decode_workspace", "aid_belief_update_workspace",
                 "aid_elbo_step_workspace", "aid_efe_step_workspace", "aid_value_step_workspace", "aid_dynamics_step_workspace",
                 "aid_small_net_workspace", "aid_encoder_saved_bytes", "aid_encoder_workspace", "aid_mine_train_workspace",
                 "aid_conv_decoder_workspace", "aid_pixel_epistemic_workspace", "aid_epistemic_value_workspace", "aid_gemm_scratch_bytes", "aid_module_workspace", "aid_jacobian_features_workspace", "aid_decoder_block_workspace",
                 "aid_pixel_jacobian_features_workspace"):
        getattr(L, name).restype = C.c_size_t
    _lib = L
    if L.aid_version() != ABI_VERSION:
        raise RuntimeError(f"aid_amd: ABI version mismatch: aid_amd/_native.py binds version {ABI_VERSION} of include/aid_hip.h, "
                           f"{path} reports {L.aid_version()}; rebuild the library")
    return L


ABI_VERSION = 2        # AID_ABI_VERSION of include/aid_hip.h this module's ctypes signatures were written against


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().aid_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "native ops take contiguous CUDA float32 tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Workspace:
    """Caller-owned scratch buffer handed to the native ops (grown on demand, never shrunk)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=self.device)
        return self.buf


@_pins_params
def score_net_table(params: Dict[str, torch.Tensor], prefix: str, L: int, O: int, D: int, n_blocks: int = 6,
                    TE: int = 128) -> AidScoreNet:
    """Build the AidScoreNet pointer table from tensors keyed by reference state_dict names."""
    t = AidScoreNet()
    t.L, t.O, t.D, t.TE, t.n_blocks = L, O, D, TE, n_blocks
    for key, field in SCORE_KEYMAP.items():
        setattr(t, field, _ptr(params[prefix + key]).value)
    for i in range(n_blocks):
        for key, field in BLOCK_KEYMAP.items():
            setattr(t.blk[i], field, _ptr(params[f"{prefix}transformer_blocks.{i}.{key}"]).value)
    return t


# ------------------------------------------------------------------------------------------------
def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, a_kmajor=0, b_kmajor=0, bias=None, act="none", residual=None,
         ldr=0, alpha=1.0, accumulate=0, scratch=None):
    if scratch is not None:      # with scratch the mid-size split-K route (K slices folded in the kernel) is open to the dispatcher
        check(lib().aid_gemm_f32_scratch(_ptr(A), lda, a_kmajor, _ptr(B), ldb, b_kmajor, _ptr(C_out), ldc, M, N, K, _ptr(bias), ACT[act],
                                         _ptr(residual), ldr, C.c_float(alpha), accumulate, _ptr(scratch), C.c_size_t(scratch.numel() * scratch.element_size()),
                                         _stream()), "aid_gemm_f32_scratch")
        return C_out
    check(lib().aid_gemm_f32(_ptr(A), lda, a_kmajor, _ptr(B), ldb, b_kmajor, _ptr(C_out), ldc, M, N, K, _ptr(bias), ACT[act],
                             _ptr(residual), ldr, C.c_float(alpha), accumulate, _stream()), "aid_gemm_f32")
    return C_out


def gemm_emulation(mode: int):
    """aid_gemm_emulation: 0 = the tall full-row GEMMs (the estimator's layers, the reverse loop's hoisted conditioning) on the float32 MFMA pipe (default);
    1 = on the bf16 pipe as six split products at float32 accuracy (csrc/gemm_rowln_x6.hip).  Process-wide."""
    check(lib().aid_gemm_emulation(int(mode)), "aid_gemm_emulation")


def gemm_emulation_mode() -> int:
    return int(lib().aid_gemm_emulation_mode())


def gemm_scratch(device="cuda"):
    """Scratch for gemm(..., scratch=): K slices and tile counters of the mid-size split-K route."""
    return torch.empty((lib().aid_gemm_scratch_bytes() + 3) // 4, dtype=torch.float32, device=device)


def fill_normal(out, seed, offset):
    check(lib().aid_fill_normal(_ptr(out), C.c_size_t(out.numel()), C.c_uint64(seed), C.c_uint64(offset), _stream()), "aid_fill_normal")


def fill_uniform(out, seed, offset):
    check(lib().aid_fill_uniform(_ptr(out), C.c_size_t(out.numel()), C.c_uint64(seed), C.c_uint64(offset), _stream()), "aid_fill_uniform")


def fill_dropout_mask(out, p, seed, offset):
    check(lib().aid_fill_dropout_mask(_ptr(out), C.c_size_t(out.numel()), C.c_float(p), C.c_uint64(seed), C.c_uint64(offset), _stream()),
          "aid_fill_dropout_mask")


def score_forward(table: AidScoreNet, z_t, t, obs, continuous: bool, drop_mask=None, ws: Workspace = None):
    B = z_t.shape[0]
    out = torch.empty(B, table.L, device=z_t.device, dtype=torch.float32)
    nbytes = lib().aid_score_forward_workspace(C.byref(table), B)
    ws = ws or Workspace(z_t.device)
    buf = ws.get(nbytes)
    check(lib().aid_score_forward(C.byref(table), B, _ptr(z_t), _ptr(t), T_CONTINUOUS if continuous else T_DISCRETE, _ptr(obs),
                                  _ptr(drop_mask), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_score_forward")
    return out


def conv3_igemm(W, x_cnhw, stride: int):
    """3x3 conv, pad 1, as an implicit GEMM.  W [Cout, ldw] (ldw >= 9 Cin, % 4 == 0), x [Cin, B, Hi, Hi] -> [Cout, B, Ho, Ho]"""
    Cin, B, Hi, _ = x_cnhw.shape; Cout = W.shape[0]
    Ho = (Hi + 2 - 3) // stride + 1
    out = torch.empty(Cout, B, Ho, Ho, device=W.device, dtype=torch.float32)
    check(lib().aid_conv3_igemm(_ptr(W), W.shape[1], _ptr(x_cnhw), _ptr(out), Cin, Cout, B, Hi, Ho, stride, _stream()), "aid_conv3_igemm")
    return out


def conv3_igemm_tm(W, x_cnhw, stride: int, flip_transpose: bool = False):
    """The same through the tap-major kernels: the weights are re-ordered (aid_conv3_tapmajor_weights) and aid_conv3_igemm_tm runs.  flip_transpose: W is the
    FORWARD weight [Cout_fwd, ldw] of a stride-1 conv and x is dy [Cout_fwd, B, H, H]: the result is the data gradient [Cin_fwd, B, H, H] (Cin_fwd = ldw-implied)."""
    if not flip_transpose:
        Cin, B, Hi, _ = x_cnhw.shape; Cout = W.shape[0]
        Wtm = torch.empty_like(W)
        check(lib().aid_conv3_tapmajor_weights(_ptr(W), W.shape[1], _ptr(Wtm), W.shape[1], Cin, Cout, 0, _stream()), "aid_conv3_tapmajor_weights")
    else:
        Cout_f, B, Hi, _ = x_cnhw.shape; Cin_f = W.shape[1] // 9
        ldwt = (Cout_f * 9 + 3) // 4 * 4
        Wtm = torch.empty(Cin_f, ldwt, device=W.device, dtype=torch.float32)
        check(lib().aid_conv3_tapmajor_weights(_ptr(W), W.shape[1], _ptr(Wtm), ldwt, Cin_f, Cout_f, 1, _stream()), "aid_conv3_tapmajor_weights")
        Cin, Cout = Cout_f, Cin_f
    Ho = (Hi + 2 - 3) // stride + 1
    out = torch.empty(Cout, B, Ho, Ho, device=W.device, dtype=torch.float32)
    check(lib().aid_conv3_igemm_tm(_ptr(Wtm), Wtm.shape[1], _ptr(x_cnhw), _ptr(out), Cin, Cout, B, Hi, Ho, stride, _stream()), "aid_conv3_igemm_tm")
    return out


def conv3_flip_weights(W, Cin: int):
    Cout = W.shape[0]; ldwt = (Cout * 9 + 3) // 4 * 4
    Wt = torch.empty(Cin, ldwt, device=W.device, dtype=torch.float32)
    check(lib().aid_conv3_flip_weights(_ptr(W), W.shape[1], _ptr(Wt), ldwt, Cin, Cout, _stream()), "aid_conv3_flip_weights")
    return Wt


def conv3_igemm_dw(dy_cnhw, x_cnhw, ldw: int, stride: int, slabs: int = 8):
    Cin, B, Hi, _ = x_cnhw.shape; Cout, _, Ho, _ = dy_cnhw.shape
    dW = torch.empty(Cout, ldw, device=x_cnhw.device, dtype=torch.float32)
    slab = torch.empty(slabs, Cout, ldw, device=x_cnhw.device, dtype=torch.float32)
    check(lib().aid_conv3_igemm_dw(_ptr(dy_cnhw), _ptr(x_cnhw), _ptr(dW), ldw, Cin, Cout, B, Hi, Ho, stride, _ptr(slab), slabs, _stream()), "aid_conv3_igemm_dw")
    return dW


def gemm_group_tn(dW, dy, x, db=None, scratch=None):
    """dW[i] += dy[i]^T x[i] (and db[i] += dy[i].sum(0)) for every i in ONE launch (csrc/gemm_group.hip).  Lists of dense row-major tensors;
    consecutive entries with the same dW tensor are contracted into the same accumulators."""
    n = len(dW)
    PP = C.c_void_p * n
    II = C.c_int * n
    for w, g, a in zip(dW, dy, x):
        assert g.shape[0] == a.shape[0] and w.shape == (g.shape[1], a.shape[1])
    check(lib().aid_gemm_group_tn(n, PP(*[_ptr(t).value for t in dW]), PP(*[(_ptr(t).value or 0) for t in db]) if db is not None else None,
                                  PP(*[_ptr(t).value for t in dy]), PP(*[_ptr(t).value for t in x]), II(*[g.shape[0] for g in dy]),
                                  II(*[w.shape[0] for w in dW]), II(*[w.shape[1] for w in dW]), _ptr(scratch),
                                  C.c_size_t(scratch.numel() if scratch is not None else 0), _stream()), "aid_gemm_group_tn")


def gradient_penalty(table: AidScoreNet, z, t, obs, continuous: bool, drop_mask=None, ws: Workspace = None):
    """-> (penalty device scalar, g [B,L] = d(sum score)/dz)"""
    B = z.shape[0]
    pen = torch.empty((), device=z.device, dtype=torch.float32)
    g = torch.empty(B, table.L, device=z.device, dtype=torch.float32)
    ws = ws or Workspace(z.device)
    buf = ws.get(lib().aid_elbo_step_workspace(C.byref(table), None, B))
    check(lib().aid_gradient_penalty(C.byref(table), B, _ptr(z), _ptr(t), T_CONTINUOUS if continuous else T_DISCRETE, _ptr(obs), _ptr(drop_mask),
                                     _ptr(pen), _ptr(g), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_gradient_penalty")
    return pen, g


def score_backward(table: AidScoreNet, gtable: AidScoreNet, z, t, obs, d_score, continuous: bool, drop_mask=None, d_obs=None, ws: Workspace = None):
    """Forward + reverse pass of the score network for the cotangent d_score: parameter gradients accumulated into `gtable`'s tensors.
    -> (score [B,L], dz [B,L]); d_obs (optional [B,O]) accumulates dLoss/dobs."""
    B = z.shape[0]
    score = torch.empty(B, table.L, device=z.device, dtype=torch.float32)
    dz = torch.empty_like(score)
    ws = ws or Workspace(z.device)
    buf = ws.get(lib().aid_elbo_step_workspace(C.byref(table), None, B))
    check(lib().aid_score_backward(C.byref(table), C.byref(gtable), B, _ptr(z), _ptr(t), T_CONTINUOUS if continuous else T_DISCRETE, _ptr(obs), _ptr(drop_mask),
                                   _ptr(d_score), _ptr(score), _ptr(dz), _ptr(d_obs), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_score_backward")
    return score, dz


def mse(a, b):
    """F.mse_loss over two [M,N] tensors -> device scalar"""
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    out = torch.empty((), device=a.device, dtype=torch.float32)
    check(lib().aid_mse(_ptr(a2), a2.shape[1], _ptr(b2), b2.shape[1], a2.shape[0], a2.shape[1], _ptr(out), _stream()), "aid_mse")
    return out


def denoise_loop(table: AidScoreNet, coefs, obs, n_steps: int, z_init=None, step_noise=None, drop_masks=None, train=True,
                 seed=0, offset=0, keep_traj=False, ws: Workspace = None):
    B = obs.shape[0]
    out = torch.empty(B, table.L, device=obs.device, dtype=torch.float32)
    traj = torch.empty(n_steps + 1, B, table.L, device=obs.device, dtype=torch.float32) if keep_traj else None
    nbytes = lib().aid_denoise_loop_workspace(C.byref(table), B, n_steps)
    ws = ws or Workspace(obs.device)
    buf = ws.get(nbytes)
    check(lib().aid_denoise_loop(C.byref(table), B, n_steps, _ptr(coefs), _ptr(obs), _ptr(z_init), _ptr(step_noise), _ptr(drop_masks),
                                 int(bool(train)), C.c_uint64(seed), C.c_uint64(offset), _ptr(out), _ptr(traj),
                                 C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_denoise_loop")
    return (out, traj) if keep_traj else out


def prof_enable(on: bool):
    check(lib().aid_prof_enable(int(on)), "aid_prof_enable")


def prof_reset():
    check(lib().aid_prof_reset(), "aid_prof_reset")


def prof_read(kernel_id: int):
    out = (C.c_double * 4)()
    check(lib().aid_prof_read(kernel_id, out), "aid_prof_read")
    return {"launches": int(out[0]), "ms": out[1], "flops": out[2], "bytes": out[3]}


def prof_dump(path: str):
    check(lib().aid_prof_dump(path.encode()), "aid_prof_dump")


def decode_observation(dec: AidDecoderNet, z, masks=None, ws: Workspace = None):
    B = z.shape[0]
    out = torch.empty(B, dec.O, device=z.device, dtype=torch.float32)
    ws = ws or Workspace(z.device)
    buf = ws.get(lib().aid_decode_workspace(C.byref(dec), B))
    m = masks or (None, None, None)
    check(lib().aid_decode_observation(C.byref(dec), B, _ptr(z), _ptr(m[0]), _ptr(m[1]), _ptr(m[2]), _ptr(out), C.c_void_p(buf.data_ptr()),
                                       C.c_size_t(buf.numel()), _stream()), "aid_decode_observation")
    return out


def belief_update(table: AidScoreNet, dec: AidDecoderNet, coefs, obs, n_steps, z_init=None, step_noise=None, drop_masks=None,
                  dec_masks=None, train=True, seed=0, offset=0, ws: Workspace = None):
    """-> (latent [B,L], stats [2L+1] = mean, std, reconstruction error)"""
    B = obs.shape[0]
    out = torch.empty(B, table.L, device=obs.device, dtype=torch.float32)
    stats = torch.empty(2 * table.L + 1, device=obs.device, dtype=torch.float32)
    ws = ws or Workspace(obs.device)
    buf = ws.get(lib().aid_belief_update_workspace(C.byref(table), C.byref(dec), B, n_steps))
    dm = (C.c_void_p * 3)(*[_ptr(m).value for m in dec_masks]) if dec_masks is not None else None
    check(lib().aid_belief_update(C.byref(table), C.byref(dec), B, n_steps, _ptr(coefs), _ptr(obs), _ptr(z_init), _ptr(step_noise),
                                  _ptr(drop_masks), dm, int(bool(train)), C.c_uint64(seed), C.c_uint64(offset), _ptr(out), _ptr(stats),
                                  C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_belief_update")
    return out, stats


ELBO_METRICS = ("reconstruction_loss", "kl_loss", "score_matching_loss", "elbo", "reward_loss", "grad_penalty", "mean_time",
                "loss_weight_mean", "loss")


def elbo_step(net: AidScoreNet, grads: AidScoreNet, dp: AidDiffusionParams, dgrads: AidDiffusionParams, dec: AidDecoderNet,
              rew: AidRewardNet, obs, rewards, latents, time_importance, first_call: bool, noise: dict = None, train=True,
              kl_weight=0.1, diffusion_weight=1.0, reward_weight=0.5, gp_weight=0.1, seed=0, offset=0, ws: Workspace = None,
              defer_time_importance=False, recon_preset=None):
    """compute_diffusion_elbo forward+backward; gradients accumulate into the `grads`/`dgrads` tables.
    -> (metrics [16] device tensor, t [B], per_sample_losses [B])"""
    B = obs.shape[0]
    nz = AidElboNoise()
    noise = noise or {}
    for k, _ in AidElboNoise._fields_:
        setattr(nz, k, _ptr(noise.get(k)).value)
    cfg = AidElboCfg(kl_weight, diffusion_weight, reward_weight, gp_weight, int(bool(train)), int(bool(first_call)), seed, offset,
                     int(bool(defer_time_importance)), 0)
    metrics = torch.zeros(16, device=obs.device, dtype=torch.float32)
    if recon_preset is not None:
        metrics[0:1].copy_(recon_preset.reshape(1))
    t_out = torch.empty(B, device=obs.device, dtype=torch.float32)
    per = torch.empty(B, device=obs.device, dtype=torch.float32)
    ws = ws or Workspace(obs.device)
    decp = C.byref(dec) if dec is not None else None
    buf = ws.get(lib().aid_elbo_step_workspace(C.byref(net), decp, B))
    check(lib().aid_elbo_step(C.byref(net), C.byref(grads), C.byref(dp), C.byref(dgrads), decp, C.byref(rew), B, _ptr(obs),
                              _ptr(rewards), _ptr(latents), C.byref(nz), C.byref(cfg), _ptr(time_importance), _ptr(t_out), _ptr(per),
                              _ptr(metrics), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_elbo_step")
    return metrics, t_out, per


def adamw_step(p, g, m, v, ema, lr, weight_decay, step, clip_norm, scratch, beta1=0.9, beta2=0.999, eps=1e-8, ema_decay=0.9999):
    check(lib().aid_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), C.c_size_t(p.numel()), C.c_float(lr), C.c_float(weight_decay),
                               C.c_float(beta1), C.c_float(beta2), C.c_float(eps), int(step), C.c_float(clip_norm), C.c_float(ema_decay),
                               _ptr(scratch), _stream()), "aid_adamw_step")


EFE_INFO = ("policy_loss", "epistemic_mean", "pragmatic_mean", "consistency_mean", "epistemic/mi_estimate", "epistemic/joint_term",
            "epistemic/marginal_term")


def efe_step(pol, pgrads, dyn, rew, val, dec, epi, latents, running_mean, horizon=5, num_trajectories=10, num_samples=10, noise=None,
             train=True, skip_epistemic=False, epistemic_weight=0.1, pragmatic_weight=1.0, consistency_weight=0.1, discount=0.99,
             temperature=1.0, mine_alpha=0.01, seed=0, offset=0, ws: Workspace = None, next_means_out=None):
    """compute_expected_free_energy_diffusion (+ policy backward when pgrads is given).
    -> (efe [B], info [16] device tensor); running_mean (device scalar tensor) is updated in place.  next_means_out [H, K*B, L]
    (optional) receives the dynamics means of every horizon step."""
    B = latents.shape[0]
    nz = AidEfeNoise(); noise = noise or {}
    for k, _ in AidEfeNoise._fields_:
        v = noise.get(k)
        if k == "perms" and v is not None:
            assert v.dtype == torch.int32 and v.is_cuda and v.is_contiguous()
            setattr(nz, k, v.data_ptr())
        else:
            setattr(nz, k, _ptr(v).value)
    cfg = AidEfeCfg(horizon, num_trajectories, num_samples, int(bool(train)), int(bool(skip_epistemic)), 0, epistemic_weight, pragmatic_weight,
                    consistency_weight, discount, temperature, mine_alpha, seed, offset)
    efe = torch.empty(B, device=latents.device, dtype=torch.float32)
    info = torch.zeros(16, device=latents.device, dtype=torch.float32)
    ws = ws or Workspace(latents.device)
    decp = C.byref(dec) if dec is not None else None
    buf = ws.get(lib().aid_efe_step_workspace(C.byref(pol), decp, B, C.byref(cfg)))
    check(lib().aid_efe_step_ex(C.byref(pol), C.byref(pgrads) if pgrads is not None else None, C.byref(dyn), C.byref(rew), C.byref(val),
                                decp, C.byref(epi) if epi is not None else None, B, _ptr(latents), C.byref(nz), C.byref(cfg), _ptr(running_mean), _ptr(efe), _ptr(info),
                                _ptr(next_means_out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_efe_step")
    return efe, info


def lambda_returns(rewards, next_values, dones, gamma=0.99, lam=0.95, n_steps=5, exclude_immediate_rewards=False):
    out = torch.empty_like(rewards)
    check(lib().aid_lambda_returns_ex(_ptr(rewards), _ptr(next_values), _ptr(dones), rewards.shape[0], C.c_float(gamma), C.c_double(lam), n_steps,
                                      int(bool(exclude_immediate_rewards)), _ptr(out), _stream()), "aid_lambda_returns")
    return out


def value_step(val, vgrads, latents, next_latents, rewards, dones, gamma=0.99, lam=0.95, n_steps=5, ws: Workspace = None):
    """-> (loss device scalar, targets [B]); gradients accumulate into vgrads.  next_latents / rewards / dones may hold more rows than
    latents: the extra rows are a lambda-return halo (aid_value_step_ex)."""
    B, Bx = latents.shape[0], next_latents.shape[0]
    assert rewards.shape[0] == Bx and dones.shape[0] == Bx and Bx >= B
    loss = torch.zeros(1, device=latents.device); targets = torch.empty(B, device=latents.device)
    ws = ws or Workspace(latents.device)
    buf = ws.get(lib().aid_value_step_workspace(C.byref(val), B))
    check(lib().aid_value_step_ex(C.byref(val), C.byref(vgrads), B, Bx, _ptr(latents), _ptr(next_latents), _ptr(rewards), _ptr(dones), C.c_float(gamma),
                                  C.c_double(lam), n_steps, _ptr(targets), _ptr(loss), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_value_step_ex")
    return loss, targets


def sample_time(time_importance, first_call: bool, B: int, seed=0, offset=0, ws: Workspace = None):
    """_importance_sample_time alone (device Philox) -> t [B]"""
    t = torch.empty(B, device=time_importance.device, dtype=torch.float32)
    ws = ws or Workspace(time_importance.device)
    buf = ws.get(8 * B + 256)
    check(lib().aid_sample_time(_ptr(time_importance), int(bool(first_call)), B, C.c_uint64(seed), C.c_uint64(offset), _ptr(t),
                                C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_sample_time")
    return t


def dynamics_step(dyn, dgrads, latents, actions, next_latents, ws: Workspace = None):
    B = latents.shape[0]
    loss = torch.zeros(1, device=latents.device)
    ws = ws or Workspace(latents.device)
    buf = ws.get(lib().aid_dynamics_step_workspace(C.byref(dyn), B))
    check(lib().aid_dynamics_step(C.byref(dyn), C.byref(dgrads), B, _ptr(latents), _ptr(actions), _ptr(next_latents), _ptr(loss),
                                  C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_dynamics_step")
    return loss


def _small_ws(ws, device, B, D):
    ws = ws or Workspace(device)
    return ws.get(lib().aid_small_net_workspace(B, D))


def policy_forward(pol, z, eps=None, ws: Workspace = None):
    B = z.shape[0]
    action = torch.empty(B, pol.A, device=z.device); logp = torch.empty(B, device=z.device); ent = torch.empty(B, device=z.device)
    buf = _small_ws(ws, z.device, B, pol.D)
    check(lib().aid_policy_forward(C.byref(pol), B, _ptr(z), _ptr(eps), _ptr(action), _ptr(logp), _ptr(ent), C.c_void_p(buf.data_ptr()),
                                   C.c_size_t(buf.numel()), _stream()), "aid_policy_forward")
    return action, logp, ent


def value_forward(val, state, t, ws: Workspace = None):
    B = state.shape[0]
    out = torch.empty(B, 1, device=state.device)
    buf = _small_ws(ws, state.device, B, val.D)
    check(lib().aid_value_forward(C.byref(val), B, _ptr(state), _ptr(t), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_value_forward")
    return out


def predict_next_latent(dyn, z, a, ws: Workspace = None):
    B = z.shape[0]
    out = torch.empty(B, dyn.L, device=z.device)
    buf = _small_ws(ws, z.device, B, dyn.D)
    check(lib().aid_predict_next_latent(C.byref(dyn), B, _ptr(z), _ptr(a), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_predict_next_latent")
    return out


def reward_forward(rew, z, ws: Workspace = None):
    B = z.shape[0]
    out = torch.empty(B, 2, device=z.device)
    buf = _small_ws(ws, z.device, B, rew.D)
    check(lib().aid_reward_forward(C.byref(rew), B, _ptr(z), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_reward_forward")
    return out


def replay_gather(obs, next_obs, actions, rewards, dones, idx):
    B, O, A = idx.shape[0], obs.shape[1], actions.shape[1]
    assert idx.dtype == torch.int64 and idx.is_cuda
    dev = obs.device
    o, n, a = torch.empty(B, O, device=dev), torch.empty(B, O, device=dev), torch.empty(B, A, device=dev)
    r, d = torch.empty(B, device=dev), torch.empty(B, device=dev)
    if obs.dtype == torch.uint8:      # byte frame storage: widened on the fly
        assert next_obs.dtype == torch.uint8 and obs.is_cuda and obs.is_contiguous() and next_obs.is_contiguous()
        check(lib().aid_replay_gather_u8(C.c_void_p(obs.data_ptr()), C.c_void_p(next_obs.data_ptr()), _ptr(actions), _ptr(rewards), _ptr(dones),
                                         C.c_void_p(idx.data_ptr()), B, O, A, _ptr(o), _ptr(n), _ptr(a), _ptr(r), _ptr(d), _stream()), "aid_replay_gather_u8")
        return o, n, a, r, d
    check(lib().aid_replay_gather(_ptr(obs), _ptr(next_obs), _ptr(actions), _ptr(rewards), _ptr(dones), C.c_void_p(idx.data_ptr()), B, O, A,
                                  _ptr(o), _ptr(n), _ptr(a), _ptr(r), _ptr(d), _stream()), "aid_replay_gather")
    return o, n, a, r, d


def sym_eig(A):
    """Symmetric eigen-decomposition in fp64 on the device (parallel cyclic Jacobi): A [n,n] -> (w [n] unsorted, V [n,n] columns)."""
    assert A.is_cuda and A.dtype == torch.float64 and A.dim() == 2 and A.shape[0] == A.shape[1]
    n = A.shape[0]
    work = A.contiguous().clone()
    w = torch.empty(n, device=A.device, dtype=torch.float64); V = torch.empty(n, n, device=A.device, dtype=torch.float64)
    check(lib().aid_sym_eig_f64(C.c_void_p(work.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(V.data_ptr()), n, _stream()), "aid_sym_eig_f64")
    return w, V


def fokker_planck_full_step(mean, lam, V, obs=None, score=None, xi=None, dt=0.01, D=0.1, lr=0.1, noise_scale=0.01, min_var=1e-6, advance=True):
    """Full-covariance belief step on (mean [n,L], lam [n,L], V [n,L,L]) in place; returns (cov [n,L,L], precision [n,L,L], entropy [n],
    condition number [n])."""
    ts = (mean, lam, V) + ((obs, score, xi) if advance else ())
    for t in ts:
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    n, L = mean.shape
    assert lam.shape == (n, L) and V.shape == (n, L, L)
    cov = torch.empty(n, L, L, device=mean.device, dtype=torch.float64); prec = torch.empty_like(cov)
    ent = torch.empty(n, device=mean.device, dtype=torch.float64); cond = torch.empty_like(ent)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(lib().aid_fokker_planck_full_step(vp(mean), vp(lam), vp(V), vp(obs if advance else None), vp(score if advance else None),
                                            vp(xi if advance else None), n, L, C.c_double(dt), C.c_double(D), C.c_double(lr), C.c_double(noise_scale),
                                            C.c_double(min_var), int(bool(advance)), vp(cov), vp(prec), vp(ent), vp(cond), _stream()),
          "aid_fokker_planck_full_step")
    return cov, prec, ent, cond


def fokker_planck_step(mean, var, obs, score, xi, dt=0.01, D=0.1, lr=0.1, noise_scale=0.01, min_var=1e-6, max_var=10.0, want_entropy=True):
    """In-place fp64 diagonal belief step over [n, L] tensors; returns entropy [n] (or None)."""
    for t in (mean, var, obs, score, xi):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    n, L = mean.shape
    ent = torch.empty(n, device=mean.device, dtype=torch.float64) if want_entropy else None
    check(lib().aid_fokker_planck_step(C.c_void_p(mean.data_ptr()), C.c_void_p(var.data_ptr()), C.c_void_p(obs.data_ptr()),
                                       C.c_void_p(score.data_ptr()), C.c_void_p(xi.data_ptr()), n, L, C.c_double(dt), C.c_double(D), C.c_double(lr),
                                       C.c_double(noise_scale), C.c_double(min_var), C.c_double(max_var),
                                       C.c_void_p(ent.data_ptr()) if ent is not None else None, _stream()), "aid_fokker_planck_step")
    return ent


def p_sample(z, score, coefs, step: int, eps=None):
    out = torch.empty_like(z)
    check(lib().aid_p_sample(_ptr(z), _ptr(score), _ptr(eps), _ptr(coefs), int(step), z.shape[0], z.shape[1], _ptr(out), _stream()), "aid_p_sample")
    return out


# ---- sub-module forwards of LatentScoreNetwork (inference): SinusoidalPositionEmbeddings, AdaptiveLayerNorm, DiTBlock (models/score_networks.py:214-291)
def sinusoid_embed(t, freq_scale, dim: int):
    out = torch.empty(t.shape[0], dim, device=t.device)
    if t.shape[0]:
        check(lib().aid_sinusoid_embed(_ptr(t), _ptr(freq_scale), t.shape[0], int(dim), _ptr(out), _stream()), "aid_sinusoid_embed")
    return out


def adaptive_layer_norm(x, cond, mod_w, mod_b, ws: Workspace = None):
    M, D = x.shape
    y = torch.empty_like(x)
    ws = ws or Workspace(x.device)
    buf = ws.get(lib().aid_module_workspace(M, D))
    check(lib().aid_adaptive_layer_norm(_ptr(x), _ptr(cond), _ptr(mod_w), _ptr(mod_b), M, D, _ptr(y), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()),
                                        _stream()), "aid_adaptive_layer_norm")
    return y


def dit_block_forward(blk: "AidDiTBlock", x, cond, ws: Workspace = None):
    M, D = x.shape
    y = torch.empty_like(x)
    ws = ws or Workspace(x.device)
    buf = ws.get(lib().aid_module_workspace(M, D))
    check(lib().aid_dit_block_forward(C.byref(blk), M, D, _ptr(x), _ptr(cond), _ptr(y), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_dit_block_forward")
    return y


def spatial_attention_forward(x, conv_w, conv_b, temperature, ws: Workspace = None):
    """SpatialAttention.forward on an NCHW tensor (visual_encoders.py:210-225)."""
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    ws = ws or Workspace(x.device)
    buf = ws.get(4 * 3 * B * H * W + 4096)
    check(lib().aid_spatial_attention_forward(_ptr(x), _ptr(conv_w), _ptr(conv_b), _ptr(temperature), B, Cc, H, W, _ptr(out), C.c_void_p(buf.data_ptr()),
                                              C.c_size_t(buf.numel()), _stream()), "aid_spatial_attention_forward")
    return out


# ---- A5: forward-process helpers of LatentDiffusionProcess (aid/core/diffusion.py:56-104, 146-174, 239-255), one launch each
def _steps_ptr(steps):
    assert steps.is_cuda and steps.dtype == torch.int64 and steps.is_contiguous(), "diffusion steps: contiguous CUDA int64 [B]"
    return C.c_void_p(steps.data_ptr())


def log_snr(t, lmin, lmax):
    out = torch.empty_like(t)
    check(lib().aid_log_snr(_ptr(t), _ptr(lmin), _ptr(lmax), _ptr(out), C.c_size_t(t.numel()), _stream()), "aid_log_snr")
    return out


def loss_weight(t, lmin, lmax):
    out = torch.empty_like(t)
    check(lib().aid_loss_weight(_ptr(t), _ptr(lmin), _ptr(lmax), _ptr(out), C.c_size_t(t.numel()), _stream()), "aid_loss_weight")
    return out


def continuous_q_sample(z, t, eps, lmin, lmax):
    """-> (z_noisy [B,L], log_snr [B], alpha [B], sigma [B])"""
    B, L = z.shape
    zn = torch.empty_like(z)
    ls, al, sg = (torch.empty(B, device=z.device) for _ in range(3))
    if B * L == 0:          # nothing to launch (an empty tensor has no storage to point at)
        return zn, ls, al, sg
    check(lib().aid_continuous_q_sample(_ptr(z), _ptr(t), _ptr(eps), _ptr(lmin), _ptr(lmax), B, L, _ptr(zn), _ptr(ls), _ptr(al), _ptr(sg), _stream()),
          "aid_continuous_q_sample")
    return zn, ls, al, sg


def sample_latent_prior(prior_mean, prior_log_std, eps):
    out = torch.empty_like(eps)
    check(lib().aid_sample_latent_prior(_ptr(prior_mean), _ptr(prior_log_std), _ptr(eps), eps.shape[0], eps.shape[1], _ptr(out), _stream()),
          "aid_sample_latent_prior")
    return out


def _step_axpby(fn, name, x0, x1, steps, c0, c1, validate):
    B, L = x0.shape
    out = torch.empty_like(x0)
    if B * L == 0:
        return out
    flag = torch.zeros(1, dtype=torch.int32, device=x0.device) if validate else None
    check(fn(_ptr(x0), _steps_ptr(steps) if name == "aid_q_sample" else _ptr(x1), _ptr(x1) if name == "aid_q_sample" else _steps_ptr(steps),
             _ptr(c0), _ptr(c1), int(c0.numel()), B, L, _ptr(out), C.c_void_p(flag.data_ptr()) if validate else C.c_void_p(0), _stream()), name)
    if validate and int(flag.item()):
        raise IndexError(f"{name}: a diffusion step lies outside [0, {int(c0.numel())})")       # the reference's gather raises here too
    return out


def q_sample(z_start, steps, eps, sqrt_acp, sqrt_1m_acp, validate=True):
    return _step_axpby(lib().aid_q_sample, "aid_q_sample", z_start, eps, steps, sqrt_acp, sqrt_1m_acp, validate)


def posterior_mean(z_start, z_t, steps, coef1, coef2, validate=True):
    return _step_axpby(lib().aid_posterior_mean, "aid_posterior_mean", z_start, z_t, steps, coef1, coef2, validate)


def reward_normalize(rewards, state):
    """state: CUDA float64 [3] = (mean, var, count), updated in place; returns normalised rewards."""
    assert state.is_cuda and state.dtype == torch.float64 and state.numel() == 3
    out = torch.empty_like(rewards)
    check(lib().aid_reward_normalize(_ptr(rewards), rewards.shape[0], C.c_void_p(state.data_ptr()), _ptr(out), _stream()), "aid_reward_normalize")
    return out


def rowwise_sqdist_means(states, prior_mean, obs, score):
    """-> [3] device tensor: mean_b sum_l (z-mu_p)^2, (o-z)^2, score^2 (FreeEnergyComputation terms)."""
    assert states.shape == obs.shape == score.shape == prior_mean.shape, "free energy needs obs, states, score of one shape (as the reference)"
    out = torch.empty(3, device=states.device, dtype=torch.float32)
    check(lib().aid_free_energy_terms(_ptr(states), _ptr(prior_mean), _ptr(obs), _ptr(score), states.shape[0], states.shape[1], _ptr(out), _stream()),
          "aid_free_energy_terms")
    return out


def time_importance_update(time_importance, t, per_sample, first_call: bool):
    check(lib().aid_time_importance_update(_ptr(time_importance), _ptr(t), _ptr(per_sample), t.shape[0], int(bool(first_call)), _stream()),
          "aid_time_importance_update")


def reward_normalize_global(stats_src, rewards, state):
    out = torch.empty_like(rewards)
    check(lib().aid_reward_normalize_global(_ptr(stats_src), stats_src.shape[0], _ptr(rewards), rewards.shape[0], C.c_void_p(state.data_ptr()),
                                            _ptr(out), _stream()), "aid_reward_normalize_global")
    return out


def stream_hint(stream: "torch.cuda.Stream", flags: int):
    """Per-stream dispatch hint (aid_stream_hint): bit 0 = small GEMMs launched on `stream` use the 35 KB-LDS panel variant so that they
    can share CUs with a large GEMM running on another stream; bit 1 = 512-wide full-row kernels launched on `stream` use 16-deep stages (74 KB
    of LDS) so that another stream's 80 KB workgroups can share their CUs (speed only).  State lives with the stream, not the process."""
    check(lib().aid_stream_hint(C.c_void_p(stream if isinstance(stream, int) else stream.cuda_stream), int(flags)), "aid_stream_hint")


def gemm_force_kernel(k: int):
    """Kernel override for benches / tests (codes: include/aid_hip.h, aid_gemm_force_kernel): 0 auto, 1 panel32, 2 ring64, 4 tile64, 6 split-K,
    8 thin TN, 9 / 11 big (prefetch 2 / 1), 10 skinny (M <= 32), 14 panel128, 15 panel64; +256 XCD-grouped tile map.  Process-wide: tests only."""
    check(lib().aid_gemm_force_kernel(int(k)), "aid_gemm_force_kernel")


def linear_ln_act(x, W, b, gamma, beta, act="silu", add=None, mask=None):
    """y = act(LayerNorm(x W^T + b) * gamma + beta) * mask + add  (aid_linear_ln_act: LayerNorm in the GEMM epilogue; N in {512, 1024}, M >= 2048)."""
    M, K = x.shape; N = W.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    code = {"none": 0, "relu": 1, "silu": 2}[act]
    check(lib().aid_linear_ln_act(_ptr(x), x.stride(0), _ptr(W), _ptr(b), _ptr(gamma), _ptr(beta), code, _ptr(mask), _ptr(add), add.stride(0) if add is not None else N,
                                  _ptr(y), N, M, N, K, _stream()), "aid_linear_ln_act")
    return y


def gemm_mid(A, W, bias=None, act="none", residual=None, ksplit=1):
    """aid_gemm_mid: ksplit == 1 -> C [M,N] = act(A W^T + bias) + residual;  ksplit > 1 -> slabs [ksplit, M, N] of raw partial products."""
    M, K = A.shape; N = W.shape[0]
    if ksplit == 1:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
        check(lib().aid_gemm_mid(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(bias), ACT[act], _ptr(residual), residual.stride(0) if residual is not None else 0,
                                 _ptr(out), N, None, M, N, K, 1, _stream()), "aid_gemm_mid")
        return out
    slab = torch.empty(ksplit, M, N, device=A.device, dtype=torch.float32)
    check(lib().aid_gemm_mid(_ptr(A), A.stride(0), _ptr(W), W.stride(0), None, 0, None, 0, None, 0, _ptr(slab), M, N, K, ksplit, _stream()), "aid_gemm_mid")
    return slab


def gemm_w4(A, W, bias=None, act="none", ksplit=1):
    """aid_gemm_w4 (<= 128 rows): ksplit == 1 -> C [M,N] = act(A W^T + bias); else raw K-slice slabs [ksplit, M, N]."""
    M, K = A.shape; N = W.shape[0]
    if ksplit == 1:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
        check(lib().aid_gemm_w4(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(bias), ACT[act], _ptr(out), N, None, M, N, K, 1, _stream()), "aid_gemm_w4")
        return out
    slab = torch.empty(ksplit, M, N, device=A.device, dtype=torch.float32)
    check(lib().aid_gemm_w4(_ptr(A), A.stride(0), _ptr(W), W.stride(0), None, 0, None, 0, _ptr(slab), M, N, K, ksplit, _stream()), "aid_gemm_w4")
    return slab


def adaln_sum(res, slab, bias, mod, want_x=True):
    """aid_adaln_sum: x = res + bias + sum(slab, 0); y = LN(x) (1 + mod[:, :D]) + mod[:, D:] -> (x or None, y)"""
    M, D = res.shape
    x = torch.empty_like(res) if want_x else None
    y = torch.empty_like(res)
    assert mod.is_cuda and mod.dtype == torch.float32 and mod.stride(1) == 1          # a [M, 2D] column window of the wide modulation buffer
    check(lib().aid_adaln_sum(_ptr(res), D, _ptr(slab), 0 if slab is None else slab.shape[0], _ptr(bias), C.c_void_p(mod.data_ptr()), mod.stride(0), _ptr(x), D, _ptr(y), D,
                              M, D, _stream()), "aid_adaln_sum")
    return x, y


def gemm_ksplit_add(A, W, C_out, bias, ksplit):
    M, K = A.shape; N = W.shape[0]
    check(lib().aid_gemm_ksplit_add(_ptr(A), K, _ptr(W), K, _ptr(C_out), N, M, N, K, _ptr(bias), int(ksplit), _stream()), "aid_gemm_ksplit_add")


def collector_sampling(table: AidScoreNet, coefs, obs, num_steps: int, max_index: int, z_init=None, step_noise=None, drop_masks=None,
                       reinit_noise=None, train=False, seed=0, offset=0, ws: Workspace = None):
    """AsyncGPUInference._batch_diffusion_sampling. -> (z [B,L], reinit_count device int32[1])"""
    B = obs.shape[0]
    out = torch.empty(B, table.L, device=obs.device, dtype=torch.float32)
    cnt = torch.zeros(1, device=obs.device, dtype=torch.int32)
    ws = ws or Workspace(obs.device)
    buf = ws.get(lib().aid_denoise_loop_workspace(C.byref(table), B, num_steps))
    check(lib().aid_collector_sampling(C.byref(table), B, int(num_steps), int(max_index), _ptr(coefs), _ptr(obs), _ptr(z_init), _ptr(step_noise),
                                       _ptr(drop_masks), _ptr(reinit_noise), int(bool(train)), C.c_uint64(seed), C.c_uint64(offset), _ptr(out),
                                       C.c_void_p(cnt.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_collector_sampling")
    return out, cnt


@_pins_params
def encoder_table(params, feature_dim: int, in_ch: int = 9, hw: int = 84, prefix: str = "") -> AidEncoderNet:
    """Table over DrQV2Encoder state_dict tensors (or their gradient twins; missing u/v entries become NULL)."""
    t = AidEncoderNet(); t.feature_dim, t.in_ch, t.H, t.W = feature_dim, in_ch, hw, hw
    g = lambda k: _ptr(params.get(prefix + k)).value
    for i in range(4):
        t.conv_w[i], t.conv_u[i], t.conv_v[i] = g(f"convs.{i}.weight_orig"), g(f"convs.{i}.weight_u"), g(f"convs.{i}.weight_v")
        t.gn_g[i], t.gn_b[i] = g(f"norms.{i}.weight"), g(f"norms.{i}.bias")
    t.att_temperature, t.att_conv_w, t.att_conv_b = g("attention.temperature"), g("attention.spatial_conv.weight"), g("attention.spatial_conv.bias")
    t.ln_g, t.ln_b = g("ln.weight"), g("ln.bias")
    t.out0 = _lin(params, prefix + "output_layers.0"); t.out1 = _ln(params, prefix + "output_layers.1")
    t.out4 = _lin(params, prefix + "output_layers.4"); t.out5 = _ln(params, prefix + "output_layers.5")
    return t


def _i32ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def encoder_forward(enc: AidEncoderNet, x, shift_h=None, shift_w=None, pad=4, drop2d_1=None, drop2d_2=None, drop_out=None, train=False,
                    saved=None, ws: Workspace = None):
    """-> (features [B, L], saved activation buffer for encoder_backward)"""
    B = x.shape[0]
    x = x.reshape(B, enc.in_ch, enc.H, enc.W).contiguous().float()
    feat = torch.empty(B, enc.feature_dim, device=x.device, dtype=torch.float32)
    nb = lib().aid_encoder_saved_bytes(C.byref(enc), B)
    if saved is None or saved.numel() < nb:
        saved = torch.empty(nb, dtype=torch.uint8, device=x.device)
    ws = ws or Workspace(x.device)
    buf = ws.get(lib().aid_encoder_workspace(C.byref(enc), B))
    check(lib().aid_encoder_forward(C.byref(enc), B, _ptr(x), _i32ptr(shift_h), _i32ptr(shift_w), int(pad), _ptr(drop2d_1), _ptr(drop2d_2), _ptr(drop_out),
                                    int(bool(train)), _ptr(feat), C.c_void_p(saved.data_ptr()), C.c_size_t(saved.numel()),
                                    C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_encoder_forward")
    return feat, saved


def encoder_backward(enc: AidEncoderNet, grads: AidEncoderNet, d_features, saved, drop2d_1=None, drop2d_2=None, drop_out=None, train=False,
                     ws: Workspace = None, phase: int = 0):
    """phase 0: the whole backward; 1: head only (gradients of ln.* / output_layers.*); 2: attention + conv stack, continuing from phase 1 in the SAME `ws`."""
    B = d_features.shape[0]
    ws = ws or Workspace(d_features.device)
    buf = ws.get(lib().aid_encoder_workspace(C.byref(enc), B))
    check(lib().aid_encoder_backward_phase(C.byref(enc), C.byref(grads), B, _ptr(d_features.contiguous()), _ptr(drop2d_1), _ptr(drop2d_2), _ptr(drop_out),
                                           int(bool(train)), C.c_void_p(saved.data_ptr()), C.c_size_t(saved.numel()), C.c_void_p(buf.data_ptr()),
                                           C.c_size_t(buf.numel()), int(phase), _stream()), "aid_encoder_backward")


def feature_decoder_step(fd: AidFeatureDecoder, fgrads: AidFeatureDecoder, latents, target, drop_mask, d_target, ws: Workspace = None):
    """recon = mse(feature_decoder(latents), target): param grads -> fgrads, d(recon)/d(target) accumulated into d_target. -> recon (device [1])"""
    B = latents.shape[0]
    loss = torch.zeros(1, device=latents.device)
    ws = ws or Workspace(latents.device)
    buf = ws.get(lib().aid_small_net_workspace(B, fd.l0.out))
    check(lib().aid_feature_decoder_step(C.byref(fd), C.byref(fgrads), B, _ptr(latents), _ptr(target), _ptr(drop_mask), _ptr(d_target), _ptr(loss),
                                         C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_feature_decoder_step")
    return loss


def feature_decoder_forward(fd: AidFeatureDecoder, latents, drop_mask=None, ws: Workspace = None):
    B = latents.shape[0]
    out = torch.empty(B, fd.l4.out, device=latents.device, dtype=torch.float32)
    ws = ws or Workspace(latents.device)
    buf = ws.get(lib().aid_small_net_workspace(B, fd.l0.out))
    check(lib().aid_feature_decoder_forward(C.byref(fd), B, _ptr(latents), _ptr(drop_mask), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()),
                                            _stream()), "aid_feature_decoder_forward")
    return out


def infonce_step(dyn: AidDynamicsNet, latents, actions, next_enc, weight, d_next_enc, ws: Workspace = None):
    B = latents.shape[0]
    loss = torch.zeros(1, device=latents.device)
    ws = ws or Workspace(latents.device)
    buf = ws.get(lib().aid_small_net_workspace(B, dyn.D) + 16 * B * B)
    check(lib().aid_infonce_step(C.byref(dyn), B, _ptr(latents), _ptr(actions), _ptr(next_enc), C.c_float(weight), _ptr(d_next_enc), _ptr(loss),
                                 C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_infonce_step")
    return loss


def grad_sumsq(g, scratch):
    check(lib().aid_grad_sumsq(_ptr(g), C.c_size_t(g.numel()), _ptr(scratch), _stream()), "aid_grad_sumsq")


def adamw_step_ext(p, g, m, v, ema, lr, weight_decay, step, clip_norm, scratch, ext_sumsq, beta1=0.9, beta2=0.999, eps=1e-8, ema_decay=0.9999):
    check(lib().aid_adamw_step_ext(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), C.c_size_t(p.numel()), C.c_float(lr), C.c_float(weight_decay),
                                   C.c_float(beta1), C.c_float(beta2), C.c_float(eps), int(step), C.c_float(clip_norm), C.c_float(ema_decay),
                                   _ptr(scratch), _ptr(ext_sumsq), _stream()), "aid_adamw_step_ext")


def mine_train_step(dyn: AidDynamicsNet, dec: AidDecoderNet, epi: AidEpistemicNet, egrads: AidEpistemicNet, latents, actions, running_mean,
                    num_samples=5, noise=None, train=True, seed=0, offset=0, alpha=0.01, ws: Workspace = None):
    """train_epistemic_estimator forward + backward (estimator gradients accumulate into `egrads`). -> out [4] = mi, joint, marginal, loss"""
    B = latents.shape[0]
    nz = AidEfeNoise(); noise = noise or {}
    for k, _ in AidEfeNoise._fields_:
        v = noise.get(k)
        if k == "perms" and v is not None:
            assert v.dtype == torch.int32 and v.is_cuda and v.is_contiguous()
            setattr(nz, k, v.data_ptr())
        else:
            setattr(nz, k, _ptr(v).value)
    out = torch.zeros(4, device=latents.device, dtype=torch.float32)
    ws = ws or Workspace(latents.device)
    buf = ws.get(lib().aid_mine_train_workspace(C.byref(dec), C.byref(dyn), B, num_samples))
    check(lib().aid_mine_train_step(C.byref(dyn), C.byref(dec), C.byref(epi), C.byref(egrads), B, num_samples, _ptr(latents), _ptr(actions), C.byref(nz),
                                    int(bool(train)), C.c_uint64(seed), C.c_uint64(offset), C.c_float(alpha), _ptr(running_mean), _ptr(out),
                                    C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_mine_train_step")
    return out


def jacobian_features(dec, epi, z, delta=None, proj_mask=None, train=False, seed=0, offset=0, attn_mask=None, ws: Workspace = None):
    """compute_jacobian_features (active_inference.py:940-993): z [R, L] -> [R, jp4.out].  `dec` / `epi` are the state-mode tables (AidDecoderNet,
    AidEpistemicNet) or the pixel-mode ones (AidConvDecoder, AidPixelEpistemicNet); delta [4, R, L] standard-normal directions (None => device Philox)."""
    z = z.contiguous().float(); R = z.shape[0]
    out = torch.empty(R, epi.jp4.out, device=z.device, dtype=torch.float32)
    if R == 0:
        return out
    ws = ws or Workspace(z.device)
    if isinstance(epi, AidPixelEpistemicNet):
        buf = ws.get(lib().aid_pixel_jacobian_features_workspace(C.byref(dec), R))
        check(lib().aid_pixel_jacobian_features(C.byref(dec), C.byref(epi), R, _ptr(z), _ptr(delta), _ptr(proj_mask), _ptr(attn_mask), int(bool(train)),
                                                C.c_uint64(seed), C.c_uint64(offset), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
              "aid_pixel_jacobian_features")
    else:
        buf = ws.get(lib().aid_jacobian_features_workspace(C.byref(dec), R))
        check(lib().aid_jacobian_features(C.byref(dec), C.byref(epi), R, _ptr(z), _ptr(delta), _ptr(proj_mask), int(bool(train)), C.c_uint64(seed),
                                          C.c_uint64(offset), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_jacobian_features")
    return out


def epistemic_value(dec: AidDecoderNet, epi: AidEpistemicNet, next_mean, next_logvar, running_mean, num_samples=5, noise=None, train=True,
                    seed=0, offset=0, alpha=0.01, ws: Workspace = None):
    """compute_epistemic_value: one gradient-free estimator forward. -> out [4] = mi, joint, marginal term, -clamp(mi, 0)"""
    B = next_mean.shape[0]
    nz = AidEfeNoise(); noise = noise or {}
    for k, _ in AidEfeNoise._fields_:
        v = noise.get(k)
        if k == "perms" and v is not None:
            assert v.dtype == torch.int32 and v.is_cuda and v.is_contiguous()
            setattr(nz, k, v.data_ptr())
        else:
            setattr(nz, k, _ptr(v).value)
    out = torch.zeros(4, device=next_mean.device, dtype=torch.float32)
    ws = ws or Workspace(next_mean.device)
    buf = ws.get(lib().aid_epistemic_value_workspace(C.byref(dec), B, num_samples))
    check(lib().aid_epistemic_value(C.byref(dec), C.byref(epi), B, num_samples, _ptr(next_mean), _ptr(next_logvar), C.byref(nz), int(bool(train)),
                                    C.c_uint64(seed), C.c_uint64(offset), C.c_float(alpha), _ptr(running_mean), _ptr(out),
                                    C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()), "aid_epistemic_value")
    return out


def adam_l2_step(p, g, m, v, lr, weight_decay, step, clip_norm, scratch, beta1=0.9, beta2=0.999, eps=1e-8):
    check(lib().aid_adam_l2_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), C.c_size_t(p.numel()), C.c_float(lr), C.c_float(weight_decay), C.c_float(beta1),
                                 C.c_float(beta2), C.c_float(eps), int(step), C.c_float(clip_norm), _ptr(scratch), _stream()), "aid_adam_l2_step")


class AidConvDecoder(C.Structure):
    _fields_ = ([("L", C.c_int32), ("hidden", C.c_int32), ("l0", AidLinear), ("ln1", AidLN), ("l4", AidLinear), ("ln5", AidLN), ("blk", AidDecoderBlock * 4)]
                + [(n, C.c_void_p) for n in ("out0_w", "out0_b", "out3_w", "out3_b", "out6_w", "out6_b")])


@_pins_params
def conv_decoder_table(params, prefix: str = "observation_decoder.") -> AidConvDecoder:
    """Table over ConvDecoder state_dict tensors (visual_encoders.py:243-330; legacy spectral-norm keys weight_orig / weight_u / weight_v)."""
    t = AidConvDecoder()
    g = lambda k: _ptr(params[prefix + k]).value
    t.L, t.hidden = params[prefix + "latent_proj.0.weight"].shape[1], params[prefix + "latent_proj.0.weight"].shape[0] // 4
    t.l0 = _lin(params, prefix + "latent_proj.0"); t.ln1 = _ln(params, prefix + "latent_proj.1")
    t.l4 = _lin(params, prefix + "latent_proj.4"); t.ln5 = _ln(params, prefix + "latent_proj.5")

    def sn(name):
        c = AidSNConv(); c.w, c.u, c.v, c.bias = g(name + ".weight_orig"), g(name + ".weight_u"), g(name + ".weight_v"), g(name + ".bias")
        return c
    for i, up in enumerate((False, True, True, True)):
        b = f"decoder_blocks.{i}."
        t.blk[i].main0 = sn(b + "main_path.0"); t.blk[i].main1 = sn(b + ("main_path.5" if up else "main_path.4"))
        if up:
            t.blk[i].res = sn(b + "residual_path.0")
    for i in (0, 3, 6):
        setattr(t, f"out{i}_w", g(f"output_proj.{i}.weight")); setattr(t, f"out{i}_b", g(f"output_proj.{i}.bias"))
    return t


def decoder_block_forward(block, x, ws: Workspace = None):
    """DecoderBlock.forward (eval): x [R, cin, h, h] -> [R, cout, ho, ho].  `block`: the aid_amd.encoder DecoderBlock holder (main_path / residual_path convs)."""
    R, cin, h, w = x.shape
    assert h == w and cin == block.in_channels, (x.shape, block.in_channels)
    up, cout = bool(block.upsample), block.out_channels

    def sn(m):
        c = AidSNConv(); c.w, c.u, c.v, c.bias = (_ptr(t.data).value for t in (m.weight_orig, m.weight_u, m.weight_v, m.bias))
        return c
    b = AidDecoderBlock()
    b.main0 = sn(block.main_path[0]); b.main1 = sn(block.main_path[5 if up else 4])
    if not isinstance(block.residual_path, torch.nn.Identity):
        b.res = sn(block.residual_path[0])
    ho = 2 * h if up else h
    out = torch.empty(R, cout, ho, ho, device=x.device, dtype=torch.float32)
    if R == 0:
        return out
    ws = ws or Workspace(x.device)
    buf = ws.get(lib().aid_decoder_block_workspace(cin, cout, int(up), R, h))
    check(lib().aid_decoder_block_forward(C.byref(b), cin, cout, int(up), R, h, _ptr(x), _ptr(out), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), _stream()),
          "aid_decoder_block_forward")
    return out


def conv_decoder_forward(dec: AidConvDecoder, z, chunk: int = 16, ws: Workspace = None):
    """ConvDecoder.forward, eval mode: z [R, L] -> images [R, 3, 168, 168]; rows are processed `chunk` at a time (the im2col of the
    32-channel 168x168 convs is 32.5 MB per image)."""
    R = z.shape[0]
    z = z.contiguous().float()
    out = torch.empty(R, 3, 168, 168, device=z.device, dtype=torch.float32)
    ws = ws or Workspace(z.device)
    for r0 in range(0, R, chunk):
        n = min(chunk, R - r0)
        buf = ws.get(lib().aid_conv_decoder_workspace(C.byref(dec), n))
        check(lib().aid_conv_decoder_forward(C.byref(dec), n, _ptr(z[r0:r0 + n]), _ptr(out[r0:r0 + n]), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()),
                                             _stream()), "aid_conv_decoder_forward")
    return out


class AidPixelEpistemicNet(C.Structure):
    _fields_ = ([("L", C.c_int32), ("_pad", C.c_int32), ("perturbation_scale", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("pp0_w", "pp0_b", "pp2_w", "pp2_b", "pp4_w", "pp4_b", "pos", "queries", "in_proj_w", "in_proj_b")]
                + [("att_out", AidLinear), ("agg0", AidLinear), ("agg_ln", AidLN), ("agg3", AidLinear), ("jp0", AidLinear), ("jp_ln", AidLN), ("jp4", AidLinear),
                   ("lp0", AidLinear), ("lp2", AidLinear), ("mn0", AidLinear), ("mn3", AidLinear), ("mn6", AidLinear)])


@_pins_params
def pixel_epistemic_table(params, prefix: str = "epistemic_estimator.") -> AidPixelEpistemicNet:
    t = AidPixelEpistemicNet()
    g = lambda k: _ptr(params[prefix + k]).value
    t.L = params[prefix + "latent_processor.0.weight"].shape[1]
    t.perturbation_scale = g("perturbation_scale")
    for i in (0, 2, 4):
        setattr(t, f"pp{i}_w", g(f"pixel_processor.{i}.weight")); setattr(t, f"pp{i}_b", g(f"pixel_processor.{i}.bias"))
    a = "spatial_aggregator."
    t.pos, t.queries, t.in_proj_w, t.in_proj_b = g(a + "pos_encoding"), g(a + "epistemic_queries"), g(a + "attention.in_proj_weight"), g(a + "attention.in_proj_bias")
    t.att_out = _lin(params, prefix + a + "attention.out_proj"); t.agg0 = _lin(params, prefix + a + "output_proj.0")
    t.agg_ln = _ln(params, prefix + a + "output_proj.1"); t.agg3 = _lin(params, prefix + a + "output_proj.3")
    t.jp0 = _lin(params, prefix + "jacobian_projector.0"); t.jp_ln = _ln(params, prefix + "jacobian_projector.1"); t.jp4 = _lin(params, prefix + "jacobian_projector.4")
    t.lp0, t.lp2 = _lin(params, prefix + "latent_processor.0"), _lin(params, prefix + "latent_processor.2")
    t.mn0, t.mn3, t.mn6 = (_lin(params, prefix + f"mine_network.{k}") for k in (0, 3, 6))
    return t


def pixel_epistemic_value(dec: AidConvDecoder, epi: AidPixelEpistemicNet, next_mean, K: int, S: int, noise=None, train=True, seed=0, offset=0,
                          ws: Workspace = None):
    """One horizon step's pixel-mode MINE statistics for all K trajectories.  -> mine [K, 4] = joint, marginal term, t_exp, mi"""
    B = next_mean.shape[0] // K
    nz = AidEfeNoise(); noise = noise or {}
    for k, _ in AidEfeNoise._fields_:
        v = noise.get(k)
        if k == "perms" and v is not None:
            assert v.dtype == torch.int32 and v.is_cuda and v.is_contiguous()
            setattr(nz, k, v.data_ptr())
        else:
            setattr(nz, k, _ptr(v).value)
    out = torch.zeros(K, 4, device=next_mean.device, dtype=torch.float32)
    ws = ws or Workspace(next_mean.device)
    buf = ws.get(lib().aid_pixel_epistemic_workspace(C.byref(dec), K, S, B))
    check(lib().aid_pixel_epistemic_value(C.byref(dec), C.byref(epi), K, S, B, _ptr(next_mean.contiguous()), C.byref(nz), _ptr(noise.get("attn_mask")),
                                          int(bool(train)), C.c_uint64(seed), C.c_uint64(offset), _ptr(out), C.c_void_p(buf.data_ptr()),
                                          C.c_size_t(buf.numel()), _stream()), "aid_pixel_epistemic_value")
    return out


def efe_add_epistemic(mine_all, efe, info, running_mean, epistemic_weight=0.1, discount=0.99, alpha=0.01):
    H, K = mine_all.shape[0], mine_all.shape[1]
    check(lib().aid_efe_add_epistemic(_ptr(mine_all.contiguous()), H, K, efe.shape[0], C.c_float(epistemic_weight), C.c_float(discount), C.c_float(alpha),
                                      _ptr(running_mean), _ptr(efe), _ptr(info), _stream()), "aid_efe_add_epistemic")
