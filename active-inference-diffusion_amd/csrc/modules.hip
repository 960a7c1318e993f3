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

// ---- SpatialAttention.forward on an NCHW tensor (encoder/visual_encoders.py:210-225): x + x sigmoid(conv7x7([mean_c x, max_c x]) / T).  Inside the encoder the
// same arithmetic runs on the CNHW activation and its last step is folded into LayerNorm(F) (conv.hip); this is the module as a stand-alone call.
__global__ void sa_pool_nchw_kernel(const float* __restrict__ x, float* __restrict__ pool /*[B][2][HW]*/, int C, int B, int HW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    const int b = i / HW, p = i % HW;
    const float* xb = x + (size_t)b * C * HW + p;
    float s = 0.f, m = -3.4e38f;
    for (int c = 0; c < C; ++c) { const float v = xb[(size_t)c * HW]; s += v; m = fmaxf(m, v); }
    pool[((size_t)b * 2) * HW + p] = s / (float)C; pool[((size_t)b * 2 + 1) * HW + p] = m;
}
__global__ void sa_conv_kernel(const float* __restrict__ pool, const float* __restrict__ w /*[2][7][7]*/, const float* __restrict__ bias, const float* __restrict__ temp,
                               float* __restrict__ att, int B, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H * W) return;
    const int x0 = i % W, y0 = (i / W) % H, b = i / (W * H);
    float s = bias[0];
    for (int ch = 0; ch < 2; ++ch)
        for (int ky = 0; ky < 7; ++ky) {
            const int yy = y0 + ky - 3; if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 7; ++kx) { const int xx = x0 + kx - 3; if (xx < 0 || xx >= W) continue;
                s += pool[(((size_t)b * 2 + ch) * H + yy) * W + xx] * w[(ch * 7 + ky) * 7 + kx]; }
        }
    att[i] = 1.f / (1.f + expf(-s / temp[0]));
}
__global__ void sa_apply_nchw_kernel(const float* __restrict__ x, const float* __restrict__ att, float* __restrict__ out, int C, int B, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * C * HW) return;
    const int p = i % HW; const size_t b = i / ((size_t)C * HW);
    out[i] = x[i] * (1.f + att[b * HW + p]);          // x + x * attention
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

int aid_spatial_attention_forward(const float* x, const float* conv_w, const float* conv_b, const float* temperature, int B, int C, int H, int W, float* out,
                                  void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(x && conv_w && conv_b && temperature && out, AID_E_NULL, "aid_spatial_attention_forward: null argument");
    AID_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AID_E_SHAPE, "aid_spatial_attention_forward: B=%d C=%d H=%d W=%d", B, C, H, W);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const int HW = H * W;
    float* pool = ws.f32((size_t)B * 2 * HW); float* att = ws.f32((size_t)B * HW); AID_WS_OK(ws);
    const unsigned gp = (unsigned)(((size_t)B * HW + 255) / 256);
    hipLaunchKernelGGL(sa_pool_nchw_kernel, dim3(gp), dim3(256), 0, s, x, pool, C, B, HW);
    hipLaunchKernelGGL(sa_conv_kernel, dim3(gp), dim3(256), 0, s, pool, conv_w, conv_b, temperature, att, B, H, W);
    hipLaunchKernelGGL(sa_apply_nchw_kernel, dim3((unsigned)(((size_t)B * C * HW + 255) / 256)), dim3(256), 0, s, x, att, out, C, B, HW);
    AID_LAUNCH_CHECK();
    return AID_OK;
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
