// Sub-module forwards of LatentScoreNetwork as stand-alone operations (reference: models/score_networks.py:214-235 DiTBlock.forward, :257-270
// AdaptiveLayerNorm.forward, :282-291 SinusoidalPositionEmbeddings.forward).  The training and sampling paths never call these - they run the whole network
// through the fused entry points (aid_score_forward, aid_denoise_loop, aid_elbo_step) - but the reference's modules are callable, so the mirror's are too:
// each is a short composition of the same kernels the fused paths launch (sinusoid_embed, adaln, gemm_f32 with fused epilogues), inference only.
#include "aid_common.h"
#include "rowwise.h"

namespace aid {

__global__ void silu_rows_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = act_apply(x[i], ACT_SILU);
}
static int silu_rows(const float* x, float* y, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    hipLaunchKernelGGL(silu_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
// y = LN(x) (1 + scale) + shift, [scale | shift] = SiLU(cond) W^T + b      (W [2D, D])
static int adaptive_ln(const float* x, const float* cond, const float* W, const float* b, int M, int D, float* y, Workspace& ws, hipStream_t s) {
    const size_t mark = ws.mark();
    float* sc = ws.f32((size_t)M * D); float* mod = ws.f32((size_t)M * 2 * D); AID_WS_OK(ws);
    AID_TRY(silu_rows(cond, sc, (size_t)M * D, s));
    AID_TRY(linear_fwd(sc, D, W, b, mod, 2 * D, M, 2 * D, D, ACT_NONE, s));
    AID_TRY(adaln(x, D, mod, 2 * D, y, D, nullptr, M, D, s));
    ws.release(mark);
    return AID_OK;
}

}  // namespace aid

using namespace aid;

extern "C" {

int aid_sinusoid_embed(const float* t, const float* freq_scale, int M, int dim, float* out, void* stream) {
    AID_REQUIRE(t && freq_scale && out, AID_E_NULL, "aid_sinusoid_embed: null argument");
    AID_REQUIRE(M >= 0 && dim >= 4 && dim % 2 == 0, AID_E_SHAPE, "aid_sinusoid_embed: M=%d dim=%d", M, dim);
    if (M == 0) return AID_OK;
    return sinusoid_embed(t, freq_scale, 1.0f, AID_T_DISCRETE, out, M, dim, (hipStream_t)stream);
}

size_t aid_module_workspace(int M, int D) { return M > 0 && D > 0 ? sizeof(float) * (size_t)M * (size_t)D * 12 + 8192 : 0; }

int aid_adaptive_layer_norm(const float* x, const float* cond, const float* mod_w, const float* mod_b, int M, int D, float* y, void* wsp, size_t ws_bytes,
                            void* stream) {
    AID_REQUIRE(x && cond && mod_w && mod_b && y, AID_E_NULL, "aid_adaptive_layer_norm: null argument");
    AID_REQUIRE(M > 0 && D > 0 && D % 4 == 0, AID_E_SHAPE, "aid_adaptive_layer_norm: M=%d D=%d", M, D);
    Workspace ws(wsp, ws_bytes);
    return adaptive_ln(x, cond, mod_w, mod_b, M, D, y, ws, (hipStream_t)stream);
}

int aid_dit_block_forward(const AidDiTBlock* blk, int M, int D, const float* x, const float* cond, float* y, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(blk && x && cond && y, AID_E_NULL, "aid_dit_block_forward: null argument");
    AID_REQUIRE(M > 0 && D > 0 && D % 4 == 0, AID_E_SHAPE, "aid_dit_block_forward: M=%d D=%d", M, D);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    float* nx = ws.f32((size_t)M * D); float* v = ws.f32((size_t)M * D); float* h = ws.f32((size_t)M * D); float* u = ws.f32((size_t)M * 4 * D);
    AID_WS_OK(ws);
    // h = x + out_proj(V_proj(adaLN1(x, cond))): a single token attends to itself with weight 1, so MultiheadAttention(q, k, v) = out_proj(V_proj(x))
    AID_TRY(adaptive_ln(x, cond, blk->ada1_w, blk->ada1_b, M, D, nx, ws, s));
    AID_TRY(linear_fwd(nx, D, blk->in_proj_w + (size_t)2 * D * D, blk->in_proj_b + 2 * D, v, D, M, D, D, ACT_NONE, s));
    AID_TRY(linear_fwd(v, D, blk->out_proj_w, blk->out_proj_b, h, D, M, D, D, ACT_NONE, s, nullptr, x, D));
    // y = h + W2 GELU(W1 adaLN2(h, cond) + b1) + b2
    AID_TRY(adaptive_ln(h, cond, blk->ada2_w, blk->ada2_b, M, D, nx, ws, s));
    AID_TRY(linear_fwd(nx, D, blk->mlp0_w, blk->mlp0_b, u, 4 * D, M, 4 * D, D, ACT_GELU, s));
    AID_TRY(linear_fwd(u, 4 * D, blk->mlp2_w, blk->mlp2_b, y, D, M, D, 4 * D, ACT_NONE, s, nullptr, h, D));
    return AID_OK;
}

}  // extern "C"
