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
ACT = {"none": 0, "relu": 1, "silu": 2, "gelu": 3}
PROF_GEMM, PROF_ROWWISE, PROF_OPTIM = 0, 1, 2

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

_lib = None


def lib() -> C.CDLL:
    """Load libaid_hip.so once; fail loudly (no fallback) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"aid_amd: native library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C active-inference-diffusion_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.aid_last_error.restype = C.c_char_p
    L.aid_version.restype = C.c_int
    for name in ("aid_score_forward_workspace", "aid_denoise_loop_workspace"):
        getattr(L, name).restype = C.c_size_t
    _lib = L
    if L.aid_version() != 1:
        raise RuntimeError("aid_amd: ABI version mismatch between aid_amd/_native.py and libaid_hip.so")
    return L


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
         ldr=0, alpha=1.0, accumulate=0):
    check(lib().aid_gemm_f32(_ptr(A), lda, a_kmajor, _ptr(B), ldb, b_kmajor, _ptr(C_out), ldc, M, N, K, _ptr(bias), ACT[act],
                             _ptr(residual), ldr, C.c_float(alpha), accumulate, _stream()), "aid_gemm_f32")
    return C_out


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
