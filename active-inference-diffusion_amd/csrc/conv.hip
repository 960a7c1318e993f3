// DrQ-v2 pixel encoder on gfx950: forward and backward (aid/encoder/visual_encoders.py:13-225, 471-492).
//
// Layout: activations live as C-N-HW ("CNHW": [channel][sample][h][w]) between the convolutions, so that
//   * conv3x3 = ONE fp32-MFMA GEMM  out[Cout][N*HW] = W_eff[Cout][Cin*9] x X[Cin*9][N*HW]  with the weight in torch's own flatten order (no
//     permutation) and X gathered from the planes by the kernel's B-operand loader (coalesced along w);
//   * a channel plane [N][HW] is contiguous: GroupNorm / per-channel reductions / spatial pooling stream it at HBM rate.
// The 3x3 convolutions are implicit GEMMs (conv_igemm.hip: forward, data gradient through the flipped weights, weight gradient): no im2col /
// col2im buffers (round 1-2 materialised 3.8 GB of them per forward and kept them for the backward).
// The 451 584-wide LayerNorm + Linear use a deterministic split-K ("slab") GEMM; conv weight gradients do too.
// Spectral normalisation follows the LEGACY torch.nn.utils.spectral_norm semantics the reference uses: one power iteration per
// training forward mutating u/v, sigma = u.(W v) with u, v constant in the backward.
#include "mlp.h"
#include "score_net.h"   // philox_fill

namespace aid {

constexpr int ENC_LAYERS = 4;

__device__ __forceinline__ float wg_sum256(float v, float* sh) {   // any blockDim multiple of 64 up to 1024; result valid in thread 0.. (broadcast via sh[0])
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += sh[i];
    __syncthreads();
    return t;
}

// ---- RandomShiftAugmentation (reflect pad + per-sample crop) fused with NCHW -> CNHW ----------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }
__global__ void aug_to_cnhw_kernel(const float* __restrict__ x, const int32_t* __restrict__ sh, const int32_t* __restrict__ sw, int pad,
                                   float* __restrict__ out, int B, int C, int H, int W) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)B * C * H * W;
    if (i >= n) return;
    const int w = i % W, h = (i / W) % H, b = (i / ((size_t)W * H)) % B, c = i / ((size_t)W * H * B);
    int hs = h, ws = w;
    if (sh) { hs = reflect_idx(h + sh[b] - pad, H); ws = reflect_idx(w + sw[b] - pad, W); }
    out[i] = x[(((size_t)b * C + c) * H + hs) * W + ws];
}

// ---- legacy spectral norm: one workgroup, W [Cout][K] (K <= 1152, Cout <= 256) ------------------------------------------------
// save layout: u_used[Cout] | v_used[K] | sigma
struct SnLayer { const float* w; float* u; float* v; int Cout, K, Kp; float* w_eff; float* save; };
struct SnBatch { SnLayer l[4]; };
__device__ __forceinline__ void spectral_norm_body(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ v, int Cout, int K, int Kp,
                                                   int power_iter, int write_uv, float* __restrict__ w_eff, float* __restrict__ save) {
    __shared__ float us[256], vs[1152], t2[256], sh[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int c = tid; c < Cout; c += blockDim.x) us[c] = u[c];
    for (int k = tid; k < K; k += blockDim.x) vs[k] = v[k];
    __syncthreads();
    if (power_iter) {
        // v = normalize(W^T u)
        float loc = 0.f;
        for (int k = tid; k < K; k += blockDim.x) {
            float s = 0.f;
            for (int c = 0; c < Cout; ++c) s += w[(size_t)c * K + k] * us[c];
            vs[k] = s; loc += s * s;
        }
        const float nv = fmaxf(sqrtf(wg_sum256(loc, sh)), 1e-12f);
        for (int k = tid; k < K; k += blockDim.x) vs[k] /= nv;
        __syncthreads();
    }
    // t2 = W v  (one wave per row)
    for (int c = wave; c < Cout; c += nw) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += w[(size_t)c * K + k] * vs[k];
        s = wave_sum(s);
        if (lane == 0) t2[c] = s;
    }
    __syncthreads();
    if (power_iter) {
        float loc = 0.f;
        for (int c = tid; c < Cout; c += blockDim.x) loc += t2[c] * t2[c];
        const float nu = fmaxf(sqrtf(wg_sum256(loc, sh)), 1e-12f);
        for (int c = tid; c < Cout; c += blockDim.x) us[c] = t2[c] / nu;
        __syncthreads();
    }
    float loc = 0.f;
    for (int c = tid; c < Cout; c += blockDim.x) loc += us[c] * t2[c];
    const float sigma = wg_sum256(loc, sh);
    for (int i = tid; i < Cout * Kp; i += blockDim.x) { const int c = i / Kp, k = i % Kp; w_eff[i] = k < K ? w[(size_t)c * K + k] / sigma : 0.f; }
    for (int c = tid; c < Cout; c += blockDim.x) { save[c] = us[c]; if (write_uv) u[c] = us[c]; }
    for (int k = tid; k < K; k += blockDim.x) { save[Cout + k] = vs[k]; if (write_uv) v[k] = vs[k]; }
    if (tid == 0) save[Cout + K] = sigma;
}
// the encoder's four layers in one launch (one workgroup each: they depend on the weights only, not on each other or on the activations - launched one
// by one in front of their convolutions they were 4 x 50-68 us of one CU on the critical path of every encoder pass)
__global__ void __launch_bounds__(1024) spectral_norm4_kernel(SnBatch bt, int power_iter, int write_uv) {
    const SnLayer& L = bt.l[blockIdx.x];
    spectral_norm_body(L.w, L.u, L.v, L.Cout, L.K, L.Kp, power_iter, write_uv, L.w_eff, L.save);
}
// dW_orig += (dW_eff - <dW_eff, W_eff> u v^T) / sigma, in two launches that use the whole chip (the one-workgroup form was 107 us per layer on ONE CU, four layers per encoder backward, on the step's
// critical path): 64 partial dots in a fixed partition, then every thread adds the 64 partials in the same order and updates its element.
constexpr int SN_BWD_PARTS = 64;
__global__ void __launch_bounds__(256) sn_bwd_dot_kernel(const float* __restrict__ dw_eff, const float* __restrict__ w_eff, int Cout, int K, int Kp, float* __restrict__ part) {
    __shared__ float sh[16];
    float loc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Cout * K; i += SN_BWD_PARTS * 256) { const int c = i / K, k = i - c * K; loc += dw_eff[(size_t)c * Kp + k] * w_eff[(size_t)c * Kp + k]; }
    const float t = wg_sum256(loc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256) sn_bwd_apply_kernel(const float* __restrict__ dw_eff, const float* __restrict__ part, const float* __restrict__ save, int Cout, int K, int Kp,
                                                           float* __restrict__ dw_orig) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * K) return;
    float dot = 0.f;
#pragma unroll 8
    for (int b = 0; b < SN_BWD_PARTS; ++b) dot += part[b];
    const int c = i / K, k = i - c * K;
    dw_orig[i] += (dw_eff[(size_t)c * Kp + k] - dot * save[c] * save[Cout + k]) / save[Cout + K];
}

// ---- GroupNorm + Mish + Dropout2d, one workgroup per (sample, group); x/y CNHW [C][B][HW] -----------------------------------------
__global__ void gn_mish_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mask /*[B][C] or null*/, float* __restrict__ stats /*[B][G][2]*/, float* __restrict__ y,
                               int C, int B, int HW, int G) {
    __shared__ float sh[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G;
    const int n = cpg * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const int c = g * cpg + i / HW; s += x[((size_t)c * B + b) * HW + i % HW]; }
    const float mean = wg_sum256(s, sh) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const int c = g * cpg + i / HW; const float d = x[((size_t)c * B + b) * HW + i % HW] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wg_sum256(q, sh) / (float)n + 1e-5f);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = g * cpg + i / HW; const size_t o = ((size_t)c * B + b) * HW + i % HW;
        float v = act_apply((x[o] - mean) * rstd * gamma[c] + beta[c], ACT_MISH);
        if (mask) v *= mask[(size_t)b * C + c];
        y[o] = v;
    }
}
// backward rows: dx = rstd (g - mean g - xhat mean(g xhat)), g = dy*mask*mish'(yhat)*gamma; also writes e = dy*mask*mish'(yhat) into dy (in place)
__global__ void gn_mish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mask, const float* __restrict__ stats, float* __restrict__ dy /*in: dL/dy, out: e*/,
                                   float* __restrict__ dx, int C, int B, int HW, int G) {
    __shared__ float sh[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G;
    const int n = cpg * HW;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    float sg = 0.f, sgx = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = g * cpg + i / HW; const size_t o = ((size_t)c * B + b) * HW + i % HW;
        const float xh = (x[o] - mean) * rstd;
        float e = dy[o] * act_grad(xh * gamma[c] + beta[c], ACT_MISH);
        if (mask) e *= mask[(size_t)b * C + c];
        dy[o] = e;
        const float gg = e * gamma[c];
        sg += gg; sgx += gg * xh;
    }
    const float mg = wg_sum256(sg, sh) / (float)n, mgx = wg_sum256(sgx, sh) / (float)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = g * cpg + i / HW; const size_t o = ((size_t)c * B + b) * HW + i % HW;
        const float xh = (x[o] - mean) * rstd;
        dx[o] = rstd * (dy[o] * gamma[c] - mg - xh * mgx);
    }
}
// per-channel affine grads: dgamma[c] += sum_{b,hw} e*xhat ; dbeta[c] += sum e      (channel plane [B][HW] contiguous)
__global__ void gn_param_grad_kernel(const float* __restrict__ e, const float* __restrict__ x, const float* __restrict__ stats,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int B, int HW, int G) {
    __shared__ float sh[16];
    const int c = blockIdx.x, g = c / (C / G);
    float s0 = 0.f, s1 = 0.f;
    for (size_t i = threadIdx.x; i < (size_t)B * HW; i += blockDim.x) {
        const int b = i / HW; const size_t o = (size_t)c * B * HW + i;
        const float ee = e[o];
        s0 += ee * (x[o] - stats[2 * (b * G + g)]) * stats[2 * (b * G + g) + 1]; s1 += ee;
    }
    const float a0 = wg_sum256(s0, sh), a1 = wg_sum256(s1, sh);
    if (threadIdx.x == 0) { dgamma[c] += a0; dbeta[c] += a1; }
}

// Register-resident variants for HW % 4 == 0, HW / 4 <= 512 and CPG channels per group (the DrQ-v2 stack: 42x42 planes, 4 or 8 channels per
// group): a (sample, group) slab is CPG contiguous planes of HW floats; every thread keeps its <= 2 float4 per plane in registers, so x (and
// dy) is read once with 16-B accesses and no index division; the backward also emits the per-(sample, channel) parameter-gradient partials
// that gn_param_fold_kernel sums over the batch in order (replaces a second full pass over e and x).
typedef float gn_f4 __attribute__((ext_vector_type(4)));
template <int CPG>
__global__ void __launch_bounds__(256) gn_mish_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mask, float* __restrict__ stats, float* __restrict__ y, int C, int B,
                                                          int HW, int G) {
    __shared__ float sh[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x, HW4 = HW >> 2;
    const bool v0 = tid < HW4, v1 = tid + 256 < HW4;
    gn_f4 v[CPG][2];
    float s = 0.f;
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) {
        const gn_f4* xp = reinterpret_cast<const gn_f4*>(x + ((size_t)(g * CPG + cc) * B + b) * HW);
        v[cc][0] = v0 ? xp[tid] : (gn_f4){0.f, 0.f, 0.f, 0.f};
        v[cc][1] = v1 ? xp[tid + 256] : (gn_f4){0.f, 0.f, 0.f, 0.f};
        s += ((v[cc][0][0] + v[cc][0][1]) + (v[cc][0][2] + v[cc][0][3])) + ((v[cc][1][0] + v[cc][1][1]) + (v[cc][1][2] + v[cc][1][3]));
    }
    const float n = (float)(CPG * HW);
    const float mean = wg_sum256(s, sh) / n;
    float q = 0.f;
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h == 0 ? v0 : v1)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[cc][h][j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wg_sum256(q, sh) / n + 1e-5f);
    if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) {
        const int c = g * CPG + cc;
        const float ga = gamma[c], be = beta[c], mk = mask ? mask[(size_t)b * C + c] : 1.f;
        gn_f4* yp = reinterpret_cast<gn_f4*>(y + ((size_t)c * B + b) * HW);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!(h == 0 ? v0 : v1)) continue;
            gn_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { float t = act_apply((v[cc][h][j] - mean) * rstd * ga + be, ACT_MISH); if (mask) t *= mk; o[j] = t; }
            yp[tid + 256 * h] = o;
        }
    }
}
template <int CPG>
__global__ void __launch_bounds__(256) gn_mish_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ mask, const float* __restrict__ stats, const float* __restrict__ dy,
                                                              float* __restrict__ dx, float* __restrict__ ppart /*[B][C][2]*/, int C, int B, int HW, int G) {
    // (round 5) ONE workgroup barrier instead of 3 x (2 CPG + 2): every partial sum (the two per channel, the two of the group) is folded over the wave at once
    // and parked in LDS, wave by wave; after the one barrier the four waves' partials are added in wave order - the same order, hence the same bits, as the
    // eighteen wg_sum256 calls this replaces (the kernel was 54 barriers of latency around 1.4 GB of traffic: 3.8 TB/s).
    __shared__ float sh[(2 * CPG + 2) * 4];
    const int b = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x, HW4 = HW >> 2, lane = tid & 63, wv = tid >> 6;
    const bool v0 = tid < HW4, v1 = tid + 256 < HW4;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    gn_f4 xh[CPG][2], e[CPG][2];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) {
        const int c = g * CPG + cc;
        const size_t base = ((size_t)c * B + b) * HW;
        const gn_f4* xp = reinterpret_cast<const gn_f4*>(x + base);
        const gn_f4* dp = reinterpret_cast<const gn_f4*>(dy + base);
        const float ga = gamma[c], be = beta[c], mk = mask ? mask[(size_t)b * C + c] : 1.f;
        float pe = 0.f, pex = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = h == 0 ? v0 : v1;
            const gn_f4 xv = ok ? xp[tid + 256 * h] : (gn_f4){0.f, 0.f, 0.f, 0.f};
            const gn_f4 dv = ok ? dp[tid + 256 * h] : (gn_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xx = (xv[j] - mean) * rstd;
                float ee = ok ? dv[j] * act_grad(xx * ga + be, ACT_MISH) : 0.f;
                if (mask) ee *= mk;
                xh[cc][h][j] = xx; e[cc][h][j] = ee;
                pe += ee; pex += ee * xx;
            }
        }
        sg += pe * ga; sgx += pex * ga;
        const float wpe = wave_sum(pe), wpex = wave_sum(pex);
        if (lane == 0) { sh[(2 * cc) * 4 + wv] = wpe; sh[(2 * cc + 1) * 4 + wv] = wpex; }
    }
    {
        const float wsg = wave_sum(sg), wsgx = wave_sum(sgx);
        if (lane == 0) { sh[(2 * CPG) * 4 + wv] = wsg; sh[(2 * CPG + 1) * 4 + wv] = wsgx; }
    }
    __syncthreads();
    auto fold4 = [&](int slot) { float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += sh[slot * 4 + w];
        return t; };
    if (tid < CPG) {                                   // per-(sample, channel) partials of dgamma / dbeta
        const int c = g * CPG + tid;
        ppart[((size_t)b * C + c) * 2] = fold4(2 * tid + 1); ppart[((size_t)b * C + c) * 2 + 1] = fold4(2 * tid);
    }
    const float n = (float)(CPG * HW);
    const float mg = fold4(2 * CPG) / n, mgx = fold4(2 * CPG + 1) / n;
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) {
        const int c = g * CPG + cc;
        const float ga = gamma[c];
        gn_f4* op = reinterpret_cast<gn_f4*>(dx + ((size_t)c * B + b) * HW);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!(h == 0 ? v0 : v1)) continue;
            gn_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (e[cc][h][j] * ga - mg - xh[cc][h][j] * mgx);
            op[tid + 256 * h] = o;
        }
    }
}
// dgamma[c] += sum_b ppart[b][c][0], dbeta[c] += sum_b ppart[b][c][1], in batch order
__global__ void gn_param_fold_kernel(const float* __restrict__ ppart, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int B) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a0 = 0.f, a1 = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {          // 8 independent loads in flight, added in batch order
        float2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float2*>(ppart + ((size_t)(b + j) * C + c) * 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0 += t[j].x; a1 += t[j].y; }
    }
    for (; b < B; ++b) { a0 += ppart[((size_t)b * C + c) * 2]; a1 += ppart[((size_t)b * C + c) * 2 + 1]; }
    dgamma[c] += a0; dbeta[c] += a1;
}
static inline bool gn_reg_ok(int C, int G, int HW, const void* a, const void* b2) {
    const int cpg = C / G;
    return (cpg == 4 || cpg == 8) && HW % 4 == 0 && HW / 4 <= 512 && ((((uintptr_t)a | (uintptr_t)b2) & 15) == 0);
}

// ---- SpatialAttention (visual_encoders.py:210-225) -------------------------------------------------------------------------------
__global__ void attn_pool_kernel(const float* __restrict__ x /*CNHW*/, float* __restrict__ pool /*[B][2][HW]*/, int32_t* __restrict__ amax /*[B][HW]*/,
                                 int C, int B, int HW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    float s = 0.f, m = -3.4e38f; int am = 0;
    for (int c = 0; c < C; ++c) { const float v = x[(size_t)c * B * HW + i]; s += v; if (v > m) { m = v; am = c; } }
    const int b = i / HW, p = i % HW;
    pool[((size_t)b * 2) * HW + p] = s / (float)C; pool[((size_t)b * 2 + 1) * HW + p] = m; amax[i] = am;
}
// (round 5) the same reduction for HW % 4 == 0 and C % 32 == 0: a thread owns FOUR consecutive pixels (16-byte loads) and a QUARTER of the channels, eight loads in
// flight; the four channel quarters of a pixel quad meet in LDS and are combined in channel order (the sum in quarter order, the maximum with the strict > of the
// serial loop: the first maximal channel wins).  The one-pixel-per-thread loop above was 4-byte loads with one dependent compare chain: 2.9 TB/s.
__global__ void __launch_bounds__(256) attn_pool4_kernel(const float* __restrict__ x /*CNHW*/, float* __restrict__ pool /*[B][2][HW]*/, int32_t* __restrict__ amax /*[B][HW]*/,
                                                         int C, int B, int HW) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float ssum[4][64][4], smax[4][64][4]; __shared__ int sarg[4][64][4];
    const int ql = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const size_t nq = (size_t)B * HW / 4, q = (size_t)blockIdx.x * 64 + ql;
    const bool ok = q < nq;
    const size_t plane4 = (size_t)B * HW / 4;
    const f4* xp = reinterpret_cast<const f4*>(x) + (ok ? q : 0);
    const int c0 = cg * (C / 4), c1 = c0 + C / 4;
    f4 s = {0.f, 0.f, 0.f, 0.f}, m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f}; int am[4] = {c0, c0, c0, c0};
    for (int c = c0; c < c1; c += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(size_t)(c + u) * plane4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += v[u][j]; if (v[u][j] > m[j]) { m[j] = v[u][j]; am[j] = c + u; } }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[cg][ql][j] = s[j]; smax[cg][ql][j] = m[j]; sarg[cg][ql][j] = am[j]; }
    __syncthreads();
    if (cg != 0 || !ok) return;
    f4 ts, tm; int ta[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = ssum[0][ql][j], mm = smax[0][ql][j]; int aa = sarg[0][ql][j];
#pragma unroll
        for (int g2 = 1; g2 < 4; ++g2) { a += ssum[g2][ql][j]; if (smax[g2][ql][j] > mm) { mm = smax[g2][ql][j]; aa = sarg[g2][ql][j]; } }
        ts[j] = a / (float)C; tm[j] = mm; ta[j] = aa;
    }
    const size_t i0 = 4 * q; const int b = (int)(i0 / HW), p = (int)(i0 % HW);          // HW % 4 == 0: a quad never straddles two images
    *reinterpret_cast<f4*>(pool + ((size_t)b * 2) * HW + p) = ts;
    *reinterpret_cast<f4*>(pool + ((size_t)b * 2 + 1) * HW + p) = tm;
    typedef int i4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<i4*>(amax + i0) = (i4){ta[0], ta[1], ta[2], ta[3]};
}
__global__ void attn_conv_kernel(const float* __restrict__ pool, const float* __restrict__ w /*[2][7][7]*/, const float* __restrict__ bias,
                                 const float* __restrict__ temp, float* __restrict__ apre, float* __restrict__ att, int B, int H, int W) {
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
    apre[i] = s; att[i] = 1.f / (1.f + expf(-s / temp[0]));
}
// (the attended activation x (1 + att) is never materialised: the LayerNorm(F) kernels below form it on the fly - AttRow)
// backward A: per (b,p): datt = sum_c dout*x ; da = datt*att(1-att)/T ; tpart[i] = -datt*att(1-att)*a/T^2 ; dxa (CNHW) = dout*(1+att)
__global__ void attn_bwd_a_kernel(const float* __restrict__ dout /*[B][C][HW]*/, const float* __restrict__ x /*CNHW*/, const float* __restrict__ att,
                                  const float* __restrict__ apre, const float* __restrict__ temp, float* __restrict__ dx /*CNHW*/,
                                  float* __restrict__ da, float* __restrict__ tpart, int C, int B, int HW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    const int b = i / HW, p = i % HW;
    const float a = att[i], T = temp[0];
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = dout[((size_t)b * C + c) * HW + p];
        s += d * x[(size_t)c * B * HW + i];
        dx[(size_t)c * B * HW + i] = d * (1.f + a);
    }
    const float g = s * a * (1.f - a);
    da[i] = g / T; tpart[i] = -g * apre[i] / (T * T);
}
// backward B: dpool = conv_transpose(da, w); then route: dx[c] += dpool_avg / C  (+ dpool_max at the argmax channel)
__global__ void attn_bwd_b_kernel(const float* __restrict__ da, const float* __restrict__ w, const int32_t* __restrict__ amax, float* __restrict__ dx,
                                  int C, int B, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H * W) return;
    const int x0 = i % W, y0 = (i / W) % H, b = i / (W * H);
    float d0 = 0.f, d1 = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
        const int yy = y0 - ky + 3; if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < 7; ++kx) { const int xx = x0 - kx + 3; if (xx < 0 || xx >= W) continue;
            const float g = da[((size_t)b * H + yy) * W + xx];
            d0 += g * w[(0 * 7 + ky) * 7 + kx]; d1 += g * w[(1 * 7 + ky) * 7 + kx]; }
    }
    const float avg = d0 / (float)C; const int am = amax[i];
    for (int c = 0; c < C; ++c) dx[(size_t)c * B * H * W + i] += avg + (c == am ? d1 : 0.f);
}
// (round 5) the same for H W % 4 == 0 and C % 32 == 0: a thread owns FOUR consecutive positions (16-byte accesses) and a QUARTER of the channels, eight
// read-modify-writes in flight; the transposed 7 x 7 convolution of a position is computed once, by the channel-quarter thread with that position's number, and
// shared through LDS (the one-position-per-thread loop above was a 4-byte load-add-store chain per channel: 4.1 TB/s over 925 MB).
__global__ void __launch_bounds__(256) attn_bwd_b4_kernel(const float* __restrict__ da, const float* __restrict__ w, const int32_t* __restrict__ amax, float* __restrict__ dx,
                                                          int C, int B, int H, int W) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float savg[64][4], smx[64][4]; __shared__ int sam[64][4];
    const int ql = threadIdx.x & 63, cg = threadIdx.x >> 6, HW = H * W;
    const size_t nq = (size_t)B * HW / 4, q = (size_t)blockIdx.x * 64 + ql;
    const bool ok = q < nq;
    {
        const size_t i = 4 * (ok ? q : 0) + cg;                          // this thread's position of the quad
        const int x0 = (int)(i % W), y0 = (int)((i / W) % H), b = (int)(i / HW);
        float d0 = 0.f, d1 = 0.f;
        for (int ky = 0; ky < 7; ++ky) {
            const int yy = y0 - ky + 3; if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 7; ++kx) { const int xx = x0 - kx + 3; if (xx < 0 || xx >= W) continue;
                const float g = da[((size_t)b * H + yy) * W + xx];
                d0 += g * w[(0 * 7 + ky) * 7 + kx]; d1 += g * w[(1 * 7 + ky) * 7 + kx]; }
        }
        savg[ql][cg] = d0 / (float)C; smx[ql][cg] = d1; sam[ql][cg] = amax[i];
    }
    __syncthreads();
    if (!ok) return;
    f4 avg, mx; int am[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { avg[j] = savg[ql][j]; mx[j] = smx[ql][j]; am[j] = sam[ql][j]; }
    const size_t plane4 = (size_t)B * HW / 4;
    f4* xp = reinterpret_cast<f4*>(dx) + q;
    const int c0 = cg * (C / 4), c1 = c0 + C / 4;
    for (int c = c0; c < c1; c += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(size_t)(c + u) * plane4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[u][j] += avg[j] + (c + u == am[j] ? mx[j] : 0.f);
            xp[(size_t)(c + u) * plane4] = v[u];
        }
    }
}
// backward C: dW[2][7][7], dbias, dtemperature.  Stage 1: every workgroup reduces its slice of the B*H*W positions to 100 partial sums
// (98 taps + dbias + dtemp); stage 2: one workgroup folds the partials in a fixed order (deterministic).
constexpr int ATTN_C_PARTS = 512;
__global__ void __launch_bounds__(256) attn_bwd_c_part_kernel(const float* __restrict__ da, const float* __restrict__ pool, const float* __restrict__ tpart,
                                                              float* __restrict__ part /*[parts][100]*/, int B, int H, int W) {
    __shared__ float sh[16];
    const int n = B * H * W;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    float acc[98];
#pragma unroll
    for (int t = 0; t < 98; ++t) acc[t] = 0.f;
    float s0 = 0.f, s1 = 0.f;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += blockDim.x) {
        const int x0 = i % W, y0 = (i / W) % H, b = i / (W * H);
        const float g = da[i];
        s0 += g; s1 += tpart[i];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float* pb = pool + ((size_t)b * 2 + ch) * H * W;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int yy = y0 + ky - 3;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const int xx = x0 + kx - 3;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[(ch * 7 + ky) * 7 + kx] += g * pb[yy * W + xx];
                }
            }
        }
    }
    // (round 5) the 100 sums folded over the wave and parked per wave, ONE barrier, then added in wave order - the bits of the 100 wg_sum256 calls
    // (300 barriers) this replaces
    __shared__ float shp[100][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 98; ++t) { const float ws_ = wave_sum(acc[t]); if (lane == 0) shp[t][wv] = ws_; }
    { const float w0 = wave_sum(s0), w1 = wave_sum(s1); if (lane == 0) { shp[98][wv] = w0; shp[99][wv] = w1; } }
    __syncthreads();
    if (threadIdx.x < 100) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += shp[threadIdx.x][w];
        part[(size_t)blockIdx.x * 100 + threadIdx.x] = tot;
    }
    (void)sh;
}
// one wave per output t (grid = 100): lane l adds partials l, l + 64, ... in that order, then the 64 lane sums are folded by the fixed butterfly of wave_sum -
// deterministic, and 8 independent loads per lane instead of one thread's chain of 512 dependent ones (that single-workgroup form was 128 us of pure latency)
__global__ void __launch_bounds__(64) attn_bwd_c_fold_kernel(const float* __restrict__ part, int parts, float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dtemp) {
    const int t = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int p = lane; p < parts; p += 64) s += part[(size_t)p * 100 + t];
    s = wave_sum(s);
    if (lane == 0) { if (t < 98) dw[t] += s; else if (t == 98) dbias[0] += s; else dtemp[0] += s; }
}

// ---- LayerNorm over F = 451 584 features, one workgroup per row -----------------------------------------------------------------------
// LayerNorm over the F = 451 584 flattened conv features of a row (1.8 MB: neither registers nor LDS hold it), one 1024-thread workgroup per row.
// 16-byte accesses, four in flight per thread; the statistics in ONE pass over the row, the normalisation in a second: x is read twice and every access is
// a float4.  The two row sums (x and x^2; in the backward dy gamma and dy gamma x_hat) are ACCUMULATED IN FLOAT64: the kernels are HBM-bound, so the
// eight extra VALU operations per float4 are free, E[x^2] - mean^2 in float64 has no cancellation problem at any offset of the row (the float32 form
// shifted by x[0] did when x[0] was an outlier), and the 451 584-term sums carry no float32 rounding at all - what is left against the float32
// reference is the reference's own rounding.  The rounded float32 mean / rstd are what the row is normalised with and what the backward reads.
// F % 4 == 0 and 16-byte aligned rows (checked by the caller) for the float4 loop, else the element loop (same float64 sums: same statistics).
__device__ __forceinline__ double wg_sum_d(double v, double* sh) {      // <= 1024 threads, every thread gets the total; fixed order
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}
// The row a LayerNorm(F) kernel normalises is the spatial attention's output x (1 + att) flattened in NCHW order - element j = c HW + p of row b is
// act[(c B + b) HW + p] (1 + att[b HW + p]) with act the last conv block's CNHW activation.  (round 5) It is formed ON THE FLY from act and att in every
// kernel that needs it (forward, both backward kernels): the attn_apply kernel that used to materialise it (462 MB read + 462 MB written per pass, 0.37 ms)
// and its [B, F] buffer in the saved-activation store are gone.  Same float32 product, so the same bits as before.  HW % 4 == 0 for the float4 loops
// (a quad never straddles two channels); else the element loops.
struct AttRow {
    const float* act; const float* att; int Bn, HW;       // act + b HW: this row's first channel plane; consecutive channels are Bn HW apart
    __device__ __forceinline__ float at(size_t j) const { const size_t c = j / HW, p = j - c * HW; return act[c * (size_t)Bn * HW + p] * (1.f + att[p]); }
    typedef float f4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ f4 at4(size_t j4) const {     // quad j4 of the row (HW % 4 == 0)
        const size_t q = (size_t)(HW >> 2), c = j4 / q, pq = j4 - c * q;
        const f4 a = *reinterpret_cast<const f4*>(act + c * (size_t)Bn * HW + 4 * pq), t = *reinterpret_cast<const f4*>(att + 4 * pq);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] * (1.f + t[e]);
        return o;
    }
};
__device__ __forceinline__ AttRow att_row(const float* act, const float* att, int b, int Bn, int HW) { return AttRow{act + (size_t)b * HW, att + (size_t)b * HW, Bn, HW}; }

__global__ void __launch_bounds__(1024) bigln_kernel(const float* __restrict__ act, const float* __restrict__ att, int Bn, int HW, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stats, size_t F, int vec) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ double shd[16];
    const AttRow xr = att_row(act, att, blockIdx.x, Bn, HW);
    float* yr = y + (size_t)blockIdx.x * F;
    const size_t F4 = F / 4;
    double s = 0.0, q = 0.0;
    if (!vec) {
        for (size_t i = threadIdx.x; i < F; i += blockDim.x) { const double v = (double)xr.at(i); s += v; q += v * v; }
    } else {
        for (size_t i = threadIdx.x; i < F4; i += 4 * (size_t)blockDim.x) {
            f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const size_t j = i + u * (size_t)blockDim.x; v[u] = j < F4 ? xr.at4(j) : (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // every term goes to float64 on its own (exact squares: 24-bit x 24-bit fits 53)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double d = (double)v[u][e]; s += d; q += d * d; }
            }
        }
    }
    const double md = wg_sum_d(s, shd) / (double)F;
    const double vard = fmax(wg_sum_d(q, shd) / (double)F - md * md, 0.0);
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(vard + 1e-5));
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
    if (!vec) {
        for (size_t i = threadIdx.x; i < F; i += blockDim.x) yr[i] = (xr.at(i) - mean) * rstd * gamma[i] + beta[i];
        return;
    }
    const f4* g4 = reinterpret_cast<const f4*>(gamma); const f4* b4 = reinterpret_cast<const f4*>(beta); f4* y4 = reinterpret_cast<f4*>(yr);
    for (size_t i = threadIdx.x; i < F4; i += 2 * (size_t)blockDim.x) {
        const size_t j1 = i + blockDim.x; const bool two = j1 < F4;
        const f4 a0 = xr.at4(i), c0 = g4[i], d0 = b4[i];
        const f4 a1 = two ? xr.at4(j1) : a0, c1 = two ? g4[j1] : c0, d1 = two ? b4[j1] : d0;
        f4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] = (a0[e] - mean) * rstd * c0[e] + d0[e]; o1[e] = (a1[e] - mean) * rstd * c1[e] + d1[e]; }
        y4[i] = o0;
        if (two) y4[j1] = o1;
    }
}
__global__ void __launch_bounds__(1024) bigln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act, const float* __restrict__ att, int Bn, int HW,
                                                         const float* __restrict__ gamma, const float* __restrict__ stats, float* __restrict__ dx, size_t F, int vec) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ double shd[16];
    const AttRow xr = att_row(act, att, blockIdx.x, Bn, HW);
    const float* dyr = dy + (size_t)blockIdx.x * F; float* dxr = dx + (size_t)blockIdx.x * F;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    double sg = 0.0, sgx = 0.0;
    if (!vec) {
        for (size_t i = threadIdx.x; i < F; i += blockDim.x) { const float g = dyr[i] * gamma[i]; sg += (double)g; sgx += (double)(g * ((xr.at(i) - mean) * rstd)); }
        const float mg = (float)(wg_sum_d(sg, shd) / (double)F), mgx = (float)(wg_sum_d(sgx, shd) / (double)F);
        for (size_t i = threadIdx.x; i < F; i += blockDim.x) dxr[i] = rstd * (dyr[i] * gamma[i] - mg - (xr.at(i) - mean) * rstd * mgx);
        return;
    }
    const size_t F4 = F / 4;
    const f4* d4 = reinterpret_cast<const f4*>(dyr); const f4* g4 = reinterpret_cast<const f4*>(gamma);
    f4* o4 = reinterpret_cast<f4*>(dxr);
    for (size_t i = threadIdx.x; i < F4; i += 2 * (size_t)blockDim.x) {
        const size_t j1 = i + blockDim.x; const bool two = j1 < F4;
        const f4 a0 = xr.at4(i), c0 = g4[i], e0 = d4[i];
        const f4 a1 = two ? xr.at4(j1) : a0, c1 = two ? g4[j1] : c0, e1 = two ? d4[j1] : e0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = e0[e] * c0[e]; sg += (double)g0; sgx += (double)(g0 * ((a0[e] - mean) * rstd));
            if (two) { const float g1 = e1[e] * c1[e]; sg += (double)g1; sgx += (double)(g1 * ((a1[e] - mean) * rstd)); }
        }
    }
    const float mg = (float)(wg_sum_d(sg, shd) / (double)F), mgx = (float)(wg_sum_d(sgx, shd) / (double)F);
    for (size_t i = threadIdx.x; i < F4; i += 2 * (size_t)blockDim.x) {
        const size_t j1 = i + blockDim.x; const bool two = j1 < F4;
        const f4 a0 = xr.at4(i), c0 = g4[i], e0 = d4[i];
        const f4 a1 = two ? xr.at4(j1) : a0, c1 = two ? g4[j1] : c0, e1 = two ? d4[j1] : e0;
        f4 r0, r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r0[e] = rstd * (e0[e] * c0[e] - mg - (a0[e] - mean) * rstd * mgx);
            r1[e] = rstd * (e1[e] * c1[e] - mg - (a1[e] - mean) * rstd * mgx);
        }
        o4[i] = r0;
        if (two) o4[j1] = r1;
    }
}
__global__ void bigln_param_grad_kernel(const float* __restrict__ dy, const float* __restrict__ act, const float* __restrict__ att, int HW,
                                        const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, size_t F) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const size_t c = i / HW, p = i - c * HW;
    const float* xc = act + c * (size_t)B * HW + p;
    float s0 = 0.f, s1 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(size_t)b * F + i], x = xc[(size_t)b * HW] * (1.f + att[(size_t)b * HW + p]);
        s0 += d * (x - stats[2 * b]) * stats[2 * b + 1]; s1 += d;
    }
    dgamma[i] += s0; dbeta[i] += s1;
}

#define L1D(kern, n, s, ...)                                                                                                \
    do {                                                                                                                    \
        if ((n) > 0) { hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s, __VA_ARGS__); AID_LAUNCH_CHECK(); } \
    } while (0)

// ------------------------------------------------------------------------------------------------------------------------------------
struct EncGeom { int cin[ENC_LAYERS], cout[ENC_LAYERS], hi[ENC_LAYERS], ho[ENC_LAYERS], stride[ENC_LAYERS], K[ENC_LAYERS], Kp[ENC_LAYERS], G[ENC_LAYERS]; int HW; size_t F; };
static EncGeom enc_geom(const AidEncoderNet& e) {
    EncGeom g; int h = e.H, c = e.in_ch;
    const int ch[ENC_LAYERS] = {32, 64, 128, 256};
    for (int i = 0; i < ENC_LAYERS; ++i) {
        g.cin[i] = c; g.cout[i] = ch[i]; g.stride[i] = i == 0 ? 2 : 1; g.hi[i] = h; g.ho[i] = (h + 2 - 3) / g.stride[i] + 1;
        g.K[i] = c * 9; g.Kp[i] = (g.K[i] + 3) / 4 * 4; g.G[i] = ch[i] / 4 < 32 ? ch[i] / 4 : 32;
        h = g.ho[i]; c = ch[i];
    }
    g.HW = h * h; g.F = (size_t)256 * g.HW;
    return g;
}

// activation store that must survive from the forward to the backward (caller-owned, aid_encoder_saved_bytes)
struct EncSaved {
    float* x0;                                   // augmented input, CNHW
    float* conv_out[ENC_LAYERS]; float* act[ENC_LAYERS]; float* gn_stats[ENC_LAYERS]; float* w_eff[ENC_LAYERS]; float* sn[ENC_LAYERS];
    float *pool, *apre, *att; int32_t* amax;
    float *ln_stats, *xln;
    LlaSave s0; float* o4; float* st5; float* feat;
};
static void enc_saved_layout(const AidEncoderNet& e, int B, Workspace& ws, EncSaved& sv) {
    const EncGeom g = enc_geom(e);
    sv.x0 = ws.f32((size_t)e.in_ch * B * e.H * e.W);
    for (int i = 0; i < ENC_LAYERS; ++i) {
        const size_t n = (size_t)g.cout[i] * B * g.ho[i] * g.ho[i];
        sv.conv_out[i] = ws.f32(n); sv.act[i] = ws.f32(n); sv.gn_stats[i] = ws.f32((size_t)B * g.G[i] * 2);
        sv.w_eff[i] = ws.f32((size_t)g.cout[i] * g.Kp[i]); sv.sn[i] = ws.f32(g.cout[i] + g.K[i] + 4);
    }
    sv.pool = ws.f32((size_t)B * 2 * g.HW); sv.apre = ws.f32((size_t)B * g.HW); sv.att = ws.f32((size_t)B * g.HW);
    sv.amax = (int32_t*)ws.raw(sizeof(int32_t) * (size_t)B * g.HW);
    sv.ln_stats = ws.f32((size_t)B * 2); sv.xln = ws.f32((size_t)B * g.F);
    lla_alloc(B, e.out0.out, ws, sv.s0);
    sv.o4 = ws.f32((size_t)B * e.out4.out); sv.st5 = ws.f32((size_t)B * 2); sv.feat = ws.f32((size_t)B * e.out4.out);
}

static int conv_slabs(size_t N) { int s = (int)(N / 8192); return s < 1 ? 1 : (s > 96 ? 96 : s); }
// K slices of the encoder head's Linear(F -> 2L) forward (gemm_f32's deep-K slab route): about two workgroups per CU (128 x 128 tiles from 128 rows,
// 64 x 64 below), a multiple of 8 so that a slice's tiles share an XCD, every slice at least 1 024 deep
static int head_slabs(size_t F, int B, int N) {
    const int bm = B >= 128 ? 128 : 64, tiles = ((B + bm - 1) / bm) * ((N + bm - 1) / bm);
    int s = ((512 / tiles) + 7) / 8 * 8;
    const int cap = (int)(F / 1024) / 8 * 8;
    if (s > cap) s = cap;
    if (s > 256) s = 256;
    return s < 1 ? 1 : s;
}
// slabs of a conv weight gradient [cout][kp]: enough (tile, slab) workgroups to fill the chip (~2 000) when the gradient itself is only a few
// 64 x 64 tiles (layer 0: 32 x 81 = 2 tiles), each slab at least 1 024 columns; never more floats than 96 slabs of the widest layer
static int conv_dw_slabs(size_t N, int cout, int kp) {
    const int tiles = ((cout + 63) / 64) * ((kp + 63) / 64);
    int s = conv_slabs(N), want = 2048 / tiles, cap = (int)(N / 1024);
    if (want > s) s = want;
    if (s > cap) s = cap;
    const size_t lim = (size_t)96 * 256 * 1152 / ((size_t)cout * kp);
    if ((size_t)s > lim) s = (int)lim;
    return s < 1 ? 1 : s;
}


// =================================================================================================================================
// ConvDecoder forward, eval mode (active_inference_diffusion/encoder/visual_encoders.py:228-459): latent -> 3 x 168 x 168 image.
// Activations are CNHW like the encoder's; every 3x3 conv is an implicit GEMM (conv_igemm.hip), the 1x1 residual convs are plain GEMMs.
// Conv bias, PixelShuffle(2), InstanceNorm2d, the residual add and Mish are ONE pass over the conv output (instnorm_kernel): the
// sub-pixel shuffle is an index transform of the read, the bias of the source channel is added on the fly.
// =================================================================================================================================
// legacy spectral norm, eval mode: sigma = u . (W v) with the stored u, v; any Cout, K
__global__ void sn_rowdot_kernel(const float* __restrict__ w, const float* __restrict__ u, const float* __restrict__ v, int K, float* __restrict__ part) {
    __shared__ float sh[16];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) s += w[(size_t)c * K + k] * v[k];
    const float t = wg_sum256(s, sh);
    if (threadIdx.x == 0) part[c] = u[c] * t;
}
// tm_cin > 0: the image is written tap-major, position k' = tap tm_cin + ci holds w[c][ci 9 + tap] (3x3 convs; conv_igemm.hip, TAPM)
__device__ __forceinline__ int weight_src_k(int k, int tm_cin) { return tm_cin > 0 ? (k % tm_cin) * 9 + k / tm_cin : k; }
__global__ void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ part, int Cout, int K, int Kp, float* __restrict__ w_eff, int tm_cin) {
    __shared__ float sh[16];
    float s = 0.f;
    for (int c = threadIdx.x; c < Cout; c += blockDim.x) s += part[c];
    const float sigma = wg_sum256(s, sh);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)Cout * Kp; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i / Kp, k = i % Kp;
        w_eff[i] = k < K ? w[(size_t)c * K + weight_src_k(k, tm_cin)] / sigma : 0.f;
    }
}
__global__ void pad_weight_kernel(const float* __restrict__ w, int Cout, int K, int Kp, float* __restrict__ w_eff, int tm_cin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cout * Kp) return;
    const int c = i / Kp, k = i % Kp;
    w_eff[i] = k < K ? w[(size_t)c * K + weight_src_k(k, tm_cin)] : 0.f;
}
// y[c][r][Y][X] = act( IN( src + bias ) [+ add] ),  src = x[c][r][Y][X]  or, with shuffle, x[4c + 2(Y&1) + (X&1)][r][Y/2][X/2]
// one workgroup per (c, r); H, W are the OUTPUT sizes
__global__ void __launch_bounds__(256) instnorm_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ add,
                                                        float* __restrict__ y, int C, int R, int H, int W, int shuffle, int act) {
    __shared__ float sh[16];
    const int c = blockIdx.x / R, r = blockIdx.x % R, HW = H * W;
    auto src = [&](int i) -> float {
        if (!shuffle) return x[((size_t)c * R + r) * HW + i] + (bias ? bias[c] : 0.f);
        const int Y = i / W, X = i % W, cs = 4 * c + 2 * (Y & 1) + (X & 1), h = H >> 1, w = W >> 1;
        return x[(((size_t)cs * R + r) * h + (Y >> 1)) * w + (X >> 1)] + (bias ? bias[cs] : 0.f);
    };
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += src(i);
    const float mean = wg_sum256(s, sh) / (float)HW;
    float q = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) { const float d = src(i) - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wg_sum256(q, sh) / (float)HW + 1e-5f);
    const size_t o = ((size_t)c * R + r) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        float v = (src(i) - mean) * rstd;
        if (add) v += add[o + i];
        y[o + i] = act_apply(v, act);
    }
}
// The same with the plane in registers (HW <= 256 PER: 21^2, 42^2, 84^2, 168^2 -> PER = 2, 7, 28, 111): the conv output is read once instead of three
// times and the sub-pixel index arithmetic runs once per element.  115 MB planes at 168 x 168: the three-pass form was 15 % of a pixel-estimator call.
// (round 4) SHUF and ACT are template arguments and the PixelShuffle index is carried, not divided: with a run-time activation switch inlined 111 times and an
// integer division per element the 168 x 168 planes moved 231 MB in 150 us (1.5 TB/s): VALU-bound.  Mish = x n / (n + 2), n = e^x (e^x + 2), with
// v_exp_f32 / v_rcp_f32 (inference path: |relative error| ~ 1e-7).
__device__ __forceinline__ float mish_fast(float x) {
    if (x > 20.f) return x;
    const float e = exp_fast(x), n = e * (e + 2.f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.f);
}
// NT threads per plane: 1024 for the 84 x 84 and 168 x 168 planes, so that a thread holds 7 / 28 elements - at 256 threads the 111 elements (and their 111
// 64-bit addresses, which the compiler keeps live to have every load in flight) took all 256 VGPRs: one wave per SIMD.
template <int PER, int NT, bool SHUF, int ACT>
__global__ void __launch_bounds__(NT) instnorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ add,
                                                            float* __restrict__ y, int C, int R, int H, int W) {
    __shared__ float sh[16];
    const int c = blockIdx.x / R, r = blockIdx.x % R, HW = H * W, tid = threadIdx.x;
    float v[PER];
    if constexpr (!SHUF) {
        const float* xp = x + ((size_t)c * R + r) * HW; const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = tid + NT * k; v[k] = i < HW ? xp[i] + b : 0.f; }
    } else {
        const int h = H >> 1, w = W >> 1;
        float b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = bias ? bias[4 * c + q] : 0.f;
        int Y = tid / W, X = tid - Y * W;                    // element i = tid + NT k as (Y, X), advanced by NT per k with carries
        const int dY = NT / W, dX = NT - dY * W;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + NT * k;
            const int sub = 2 * (Y & 1) + (X & 1);
            v[k] = i < HW ? x[(((size_t)(4 * c + sub) * R + r) * h + (Y >> 1)) * w + (X >> 1)] + b4[sub] : 0.f;
            Y += dY; X += dX;
            if (X >= W) { X -= W; ++Y; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) s += v[k];
    const float mean = wg_sum256(s, sh) / (float)HW;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const float d = v[k] - mean; q += tid + NT * k < HW ? d * d : 0.f; }
    const float rstd = 1.0f / sqrtf(wg_sum256(q, sh) / (float)HW + 1e-5f);
    const size_t o = ((size_t)c * R + r) * HW;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + NT * k;
        if (i < HW) {
            float t = (v[k] - mean) * rstd;
            if (add) t += add[o + i];
            y[o + i] = ACT == ACT_MISH ? mish_fast(t) : t;
        }
    }
}
template <int PER, int NT>
static void launch_instnorm_reg(const float* x, const float* bias, const float* add, float* y, int C, int R, int H, int W, int shuffle, int act, hipStream_t s) {
    const dim3 grid(C * R);
    if (shuffle) {
        if (act == ACT_MISH) hipLaunchKernelGGL((instnorm_reg_kernel<PER, NT, true, ACT_MISH>), grid, dim3(NT), 0, s, x, bias, add, y, C, R, H, W);
        else hipLaunchKernelGGL((instnorm_reg_kernel<PER, NT, true, ACT_NONE>), grid, dim3(NT), 0, s, x, bias, add, y, C, R, H, W);
    } else {
        if (act == ACT_MISH) hipLaunchKernelGGL((instnorm_reg_kernel<PER, NT, false, ACT_MISH>), grid, dim3(NT), 0, s, x, bias, add, y, C, R, H, W);
        else hipLaunchKernelGGL((instnorm_reg_kernel<PER, NT, false, ACT_NONE>), grid, dim3(NT), 0, s, x, bias, add, y, C, R, H, W);
    }
}
static int instnorm(const float* x, const float* bias, const float* add, float* y, int C, int R, int H, int W, int shuffle, int act, hipStream_t s) {
    const int HW = H * W; const dim3 grid(C * R);
    const bool reg = (act == ACT_MISH || act == ACT_NONE) && W <= 256 && W >= 8;
    if (reg && HW <= 512) launch_instnorm_reg<2, 256>(x, bias, add, y, C, R, H, W, shuffle, act, s);
    else if (reg && HW <= 1792) launch_instnorm_reg<7, 256>(x, bias, add, y, C, R, H, W, shuffle, act, s);
    else if (reg && HW <= 7168) launch_instnorm_reg<7, 1024>(x, bias, add, y, C, R, H, W, shuffle, act, s);
    else if (reg && HW <= 28672) launch_instnorm_reg<28, 1024>(x, bias, add, y, C, R, H, W, shuffle, act, s);
    else hipLaunchKernelGGL(instnorm_kernel, grid, dim3(256), 0, s, x, bias, add, y, C, R, H, W, shuffle, act);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
// latent_proj tail: LayerNorm over F = C*HW features of a row, Mish, stored CNHW ([C][R][HW]) for the conv stack
// (round 4) two launches of R x C / 16 workgroups instead of one of R (16 rows = 16 workgroups on 256 CUs, three passes, an integer division per element: 264 us):
// partial sums of 16-channel chunks around the row's first element, then every workgroup folds the row's partials in the same order and normalises its chunk.
constexpr int BLM_CH = 16;
// (round 5) the chunk sums of x and x^2 are taken and folded in float64 (no shift by the row's first element: that form cancelled when x[0] was an
// outlier of the row); HBM-bound, the float64 adds are free.
__global__ void __launch_bounds__(256) bigln_mish_part_kernel(const float* __restrict__ x, double* __restrict__ part, int C, int HW) {
    __shared__ double shd[16];
    const int nch = C / BLM_CH, r = blockIdx.x / nch, ch = blockIdx.x % nch;
    const size_t F = (size_t)C * HW;
    const float* xc = x + (size_t)r * F + (size_t)ch * BLM_CH * HW;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < BLM_CH * HW; i += 256) { const double d = (double)xc[i]; s += d; q += d * d; }
    const double ts = wg_sum_d(s, shd), tq = wg_sum_d(q, shd);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = ts; part[2 * blockIdx.x + 1] = tq; }
}
__global__ void __launch_bounds__(256) bigln_mish_apply_kernel(const float* __restrict__ x, const double* __restrict__ part, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y, int R, int C, int HW) {
    const int nch = C / BLM_CH, r = blockIdx.x / nch, ch = blockIdx.x % nch;
    const size_t F = (size_t)C * HW;
    const float* xr = x + (size_t)r * F;
    double s = 0.0, q = 0.0;
    for (int j = 0; j < nch; ++j) { s += part[2 * (r * nch + j)]; q += part[2 * (r * nch + j) + 1]; }      // the same order in every workgroup of the row
    const double md = s / (double)F, vard = fmax(q / (double)F - md * md, 0.0);
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(vard + 1e-5));
    for (int cc = 0; cc < BLM_CH; ++cc) {
        const int c = ch * BLM_CH + cc;
        const float* xc = xr + (size_t)c * HW; const float* gc = gamma + (size_t)c * HW; const float* bc = beta + (size_t)c * HW;
        float* yc = y + ((size_t)c * R + r) * HW;
        for (int p = threadIdx.x; p < HW; p += 256) yc[p] = mish_fast((xc[p] - mean) * rstd * gc[p] + bc[p]);
    }
}
static int bigln_mish_cnhw(const float* x, const float* gamma, const float* beta, float* y, int R, int C, int HW, Workspace& ws, hipStream_t s) {
    AID_REQUIRE(C % BLM_CH == 0, AID_E_SHAPE, "bigln_mish_cnhw: %d channels", C);
    const size_t mark = ws.mark();
    double* part = reinterpret_cast<double*>(ws.f32((size_t)4 * R * (C / BLM_CH))); AID_WS_OK(ws);       // 256-B aligned: two float64 per chunk
    hipLaunchKernelGGL(bigln_mish_part_kernel, dim3(R * (C / BLM_CH)), dim3(256), 0, s, x, part, C, HW);
    hipLaunchKernelGGL(bigln_mish_apply_kernel, dim3(R * (C / BLM_CH)), dim3(256), 0, s, x, part, gamma, beta, y, R, C, HW);
    AID_LAUNCH_CHECK();
    ws.release(mark);
    return AID_OK;
}
// out[r][c][p] = sigmoid(x[c][r][p] + bias[c])      (CNHW -> NCHW)
__global__ void sigmoid_nchw_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out, int C, int R, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * R * HW) return;
    const int p = i % HW, r = (i / HW) % R, c = i / ((size_t)HW * R);
    const float v = x[i] + bias[c];
    out[((size_t)r * C + c) * HW + p] = 1.0f / (1.0f + expf(-v));
}

struct ConvW { const float* w_eff; int cin, cout, Kp; int tm; };      // tm: w_eff rows are tap-major, k' = tap cin + ci (3x3 convs whose cin allows it)
// out[Cout][R*H*W] = 3x3 conv of x (stride 1, pad 1) as an implicit GEMM (conv_igemm.hip: nothing of size 9 Cin x R H W is written)
static int conv3_fwd(const ConvW& cw, const float* x, int R, int H, float* out, hipStream_t s) {
    return conv3_igemm_ex(cw.w_eff, cw.Kp, x, out, cw.cin, cw.cout, R, H, H, 1, cw.tm, s);
}
static int conv1_fwd(const ConvW& cw, const float* x, int R, int H, float* out, hipStream_t s) {
    const size_t N = (size_t)R * H * H;
    GemmArgs ga = gemm_args(cw.w_eff, cw.Kp, x, (int)N, out, (int)N, cw.cout, (int)N, cw.cin);
    ga.b_kmajor = 1;
    return gemm_f32(ga, s);
}
static int prep_weight(const AidSNConv& c, int cin, int cout, int k, Workspace& ws, ConvW& out, hipStream_t s) {
    const int K = cin * k * k, Kp = (K + 31) / 32 * 32;
    const int tm = (k == 3 && conv3_tapmajor_ok(cin, cout)) ? 1 : 0;
    float* w_eff = ws.f32((size_t)cout * Kp); AID_WS_OK(ws);
    if (c.u) {
        float* part = ws.f32(cout); AID_WS_OK(ws);
        hipLaunchKernelGGL(sn_rowdot_kernel, dim3(cout), dim3(256), 0, s, c.w, c.u, c.v, K, part);
        hipLaunchKernelGGL(sn_scale_kernel, dim3(64), dim3(256), 0, s, c.w, part, cout, K, Kp, w_eff, tm ? cin : 0);
    } else {
        L1D(pad_weight_kernel, (size_t)cout * Kp, s, c.w, cout, K, Kp, w_eff, tm ? cin : 0);
    }
    AID_LAUNCH_CHECK();
    out.w_eff = w_eff; out.cin = cin; out.cout = cout; out.Kp = Kp; out.tm = tm;
    return AID_OK;
}
}  // namespace aid

using namespace aid;

extern "C" {

size_t aid_encoder_saved_bytes(const AidEncoderNet* enc, int B) {
    if (!enc || B <= 0) return 0;
    Workspace ws(nullptr, 0); EncSaved sv; enc_saved_layout(*enc, B, ws, sv); return ws.off + 4096;
}
size_t aid_encoder_workspace(const AidEncoderNet* enc, int B) {
    if (!enc || B <= 0) return 0;
    const EncGeom g = enc_geom(*enc);
    const size_t N = (size_t)B * g.HW;
    size_t act = 0;
    for (int i = 0; i < ENC_LAYERS; ++i) { const size_t a = (size_t)g.cout[i] * B * g.ho[i] * g.ho[i]; if (a > act) act = a; }
    // 3 activation-sized buffers, the flipped weights of one layer, <= 96 weight-gradient slabs, the slabs of the 451 584-wide Linear
    size_t n = sizeof(float) * (3 * act + 2 * (size_t)B * g.F + 4 * N + (size_t)256 * 2304 + 97 * (size_t)256 * 1152 + (size_t)256 * B * enc->out0.out + (1 << 20));
    return n + (1 << 20);
}

// forward.  x: [B, in_ch, H, W] float (frame-stacked, already in [0,1]); shift_h/shift_w: per-sample crop offsets in [0, 2*pad] or NULL
// (no augmentation); drop2d_1 [B,64], drop2d_2 [B,128], drop_out [B, 2*feature_dim]: scaled keep masks or NULL (eval / Philox)
int aid_encoder_forward(const AidEncoderNet* enc, int B, const float* x, const int32_t* shift_h, const int32_t* shift_w, int pad,
                        const float* drop2d_1, const float* drop2d_2, const float* drop_out, int train, float* features, void* saved,
                        size_t saved_bytes, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(enc && x && features && saved, AID_E_NULL, "aid_encoder_forward: null argument");
    AID_REQUIRE(B > 0 && enc->H == enc->W && enc->in_ch * 9 <= 1152, AID_E_SHAPE, "aid_encoder_forward: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    const AidEncoderNet& e = *enc;
    const EncGeom g = enc_geom(e);
    Workspace sws(saved, saved_bytes); EncSaved sv; enc_saved_layout(e, B, sws, sv); AID_WS_OK(sws);
    Workspace ws(wsp, ws_bytes);
    const float* masks[ENC_LAYERS] = {nullptr, train ? drop2d_1 : nullptr, train ? drop2d_2 : nullptr, nullptr};
    L1D(aug_to_cnhw_kernel, (size_t)B * e.in_ch * e.H * e.W, s, x, train ? shift_h : nullptr, shift_w, pad, sv.x0, B, e.in_ch, e.H, e.W);
    const float* cur = sv.x0;
    {
        SnBatch bt;
        for (int i = 0; i < ENC_LAYERS; ++i) bt.l[i] = SnLayer{e.conv_w[i], e.conv_u[i], e.conv_v[i], g.cout[i], g.K[i], g.Kp[i], sv.w_eff[i], sv.sn[i]};
        static_assert(ENC_LAYERS == 4, "spectral_norm4_kernel batches four layers");
        hipLaunchKernelGGL(spectral_norm4_kernel, dim3(ENC_LAYERS), dim3(1024), 0, s, bt, train, train);
        AID_LAUNCH_CHECK();
    }
    for (int i = 0; i < ENC_LAYERS; ++i) {
        const size_t N = (size_t)B * g.ho[i] * g.ho[i];
        const size_t mark = ws.mark();
        (void)N;
        if (conv3_tapmajor_ok(g.cin[i], g.cout[i])) {      // tap-major contraction (conv_igemm.hip, TAPM): the weights re-ordered once per pass (<= 1.2 MB)
            float* wtm = ws.f32((size_t)g.cout[i] * g.Kp[i]); AID_WS_OK(ws);
            AID_TRY(conv3_tapmajor_weights(sv.w_eff[i], g.Kp[i], wtm, g.Kp[i], g.cin[i], g.cout[i], s));
            AID_TRY(conv3_igemm_ex(wtm, g.Kp[i], cur, sv.conv_out[i], g.cin[i], g.cout[i], B, g.hi[i], g.ho[i], g.stride[i], 1, s));
        } else
            AID_TRY(conv3_igemm(sv.w_eff[i], g.Kp[i], cur, sv.conv_out[i], g.cin[i], g.cout[i], B, g.hi[i], g.ho[i], g.stride[i], s));
        {
            const int hw = g.ho[i] * g.ho[i], cpg = g.cout[i] / g.G[i];
            if (gn_reg_ok(g.cout[i], g.G[i], hw, sv.conv_out[i], sv.act[i])) {
                if (cpg == 4) hipLaunchKernelGGL(gn_mish_reg_kernel<4>, dim3(B * g.G[i]), dim3(256), 0, s, sv.conv_out[i], e.gn_g[i], e.gn_b[i], masks[i], sv.gn_stats[i], sv.act[i], g.cout[i], B, hw, g.G[i]);
                else hipLaunchKernelGGL(gn_mish_reg_kernel<8>, dim3(B * g.G[i]), dim3(256), 0, s, sv.conv_out[i], e.gn_g[i], e.gn_b[i], masks[i], sv.gn_stats[i], sv.act[i], g.cout[i], B, hw, g.G[i]);
            } else {
                hipLaunchKernelGGL(gn_mish_kernel, dim3(B * g.G[i]), dim3(256), 0, s, sv.conv_out[i], e.gn_g[i], e.gn_b[i], masks[i], sv.gn_stats[i], sv.act[i], g.cout[i], B, hw, g.G[i]);
            }
        }
        AID_LAUNCH_CHECK();
        ws.release(mark);
        cur = sv.act[i];
    }
    const int HW = g.HW, Hh = g.ho[3];
    if (HW % 4 == 0 && ((((uintptr_t)cur | (uintptr_t)sv.pool | (uintptr_t)sv.amax) & 15) == 0)) {
        const size_t nq = (size_t)B * HW / 4;
        hipLaunchKernelGGL(attn_pool4_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(256), 0, s, cur, sv.pool, sv.amax, 256, B, HW);
    } else L1D(attn_pool_kernel, B * HW, s, cur, sv.pool, sv.amax, 256, B, HW);
    L1D(attn_conv_kernel, B * HW, s, sv.pool, e.att_conv_w, e.att_conv_b, e.att_temperature, sv.apre, sv.att, B, Hh, Hh);
    {   // LayerNorm(F) over x (1 + att), formed on the fly from the CNHW activation (no attn_apply pass, no [B, F] copy)
        const int vec = HW % 4 == 0 && ((((uintptr_t)cur | (uintptr_t)sv.att | (uintptr_t)e.ln_g | (uintptr_t)e.ln_b | (uintptr_t)sv.xln) & 15) == 0);
        hipLaunchKernelGGL(bigln_kernel, dim3(B), dim3(1024), 0, s, cur, sv.att, B, HW, e.ln_g, e.ln_b, sv.xln, sv.ln_stats, g.F, vec);
    }
    AID_LAUNCH_CHECK();
    {   // Linear(F -> 2L): deterministic split-K
        const size_t mark = ws.mark();
        const int S = head_slabs(g.F, B, e.out0.out);
        GemmArgs ga = gemm_args(sv.xln, (int)g.F, e.out0.w, (int)g.F, sv.s0.lin, e.out0.out, B, e.out0.out, (int)g.F);
        ga.bias = e.out0.b; ga.slab = ws.f32((size_t)S * B * e.out0.out); ga.nslab = S; AID_WS_OK(ws);
        AID_TRY(gemm_f32(ga, s));
        ws.release(mark);
    }
    AID_TRY(ln_act(sv.s0.lin, e.out0.out, e.out1.g, e.out1.b, ACT_MISH, train ? drop_out : nullptr, sv.s0.y, e.out0.out, sv.s0.stats, B, e.out0.out, s));
    AID_TRY(linear_fwd(sv.s0.y, e.out0.out, e.out4.w, e.out4.b, sv.o4, e.out4.out, B, e.out4.out, e.out4.in, ACT_NONE, s));
    AID_TRY(ln_act(sv.o4, e.out4.out, e.out5.g, e.out5.b, ACT_TANH, nullptr, sv.feat, e.out4.out, sv.st5, B, e.out4.out, s));
    AID_TRY(copy_f32(sv.feat, features, (size_t)B * e.out4.out, s));
    return AID_OK;
}

// backward: d_features [B, L] -> parameter gradients ACCUMULATED into `grads` (same table type; conv_u/conv_v/unused members ignored)
// phase: 0 = the whole backward; 1 = the head only (output layers, Linear(F -> 2L), LayerNorm(F): everything that writes the gradients of ln.* and
// output_layers.*, ending with d(loss)/d(flattened attention output) in the workspace); 2 = the rest (spatial attention, conv stack), reading what phase 1
// left.  A data-parallel caller runs phase 1 of every pass, starts the all-reduce of the 462 MB Linear(F -> 2L) gradient, and runs phase 2 underneath it.
// Phases 1 and 2 of one pass must get the SAME workspace, untouched in between (its carve is deterministic: the hand-over buffer has the same address).
static int encoder_backward_impl(const AidEncoderNet* enc, const AidEncoderNet* grads, int B, const float* d_features, const float* drop2d_1,
                                 const float* drop2d_2, const float* drop_out, int train, const void* saved, size_t saved_bytes, void* wsp,
                                 size_t ws_bytes, int phase, void* stream) {
    AID_REQUIRE(enc && grads && (d_features || phase == 2) && saved, AID_E_NULL, "aid_encoder_backward: null argument");
    AID_REQUIRE(phase >= 0 && phase <= 2, AID_E_SHAPE, "aid_encoder_backward: phase %d", phase);
    hipStream_t s = (hipStream_t)stream;
    const AidEncoderNet& e = *enc; const AidEncoderNet& gr = *grads;
    const EncGeom g = enc_geom(e);
    Workspace sws((void*)saved, saved_bytes); EncSaved sv; enc_saved_layout(e, B, sws, sv); AID_WS_OK(sws);
    Workspace ws(wsp, ws_bytes);
    const int L = e.out4.out, L2 = e.out0.out, HW = g.HW, Hh = g.ho[3];
    const float* masks[ENC_LAYERS] = {nullptr, train ? drop2d_1 : nullptr, train ? drop2d_2 : nullptr, nullptr};
    float* d_o4 = ws.f32((size_t)B * L); float* dyeff = ws.f32((size_t)B * (L2 > L ? L2 : L)); float* dh1 = ws.f32((size_t)B * L2);
    float* d_o0 = ws.f32((size_t)B * L2);
    float* dxln = ws.f32((size_t)B * g.F); float* dxf = ws.f32((size_t)B * g.F);
    AID_WS_OK(ws);
    if (phase != 2) {
    // ---- output head: tanh(LN(Linear)) <- Dropout(Mish(LN(Linear(F))))
    AID_TRY(ln_bwd_rows(d_features, L, sv.o4, L, sv.st5, e.out5.g, e.out5.b, ACT_TANH, nullptr, d_o4, L, dyeff, B, L, s));
    AID_TRY(ln_param_grad(dyeff, sv.o4, L, sv.st5, gr.out5.g, gr.out5.b, B, L, s, &ws));
    AID_TRY(linear_bwd_weight(d_o4, L, sv.s0.y, L2, gr.out4.w, B, L, L2, s, 1));
    AID_TRY(colsum(d_o4, L, gr.out4.b, B, L, 1, s, &ws));
    AID_TRY(linear_bwd_data(d_o4, L, e.out4.w, L2, dh1, L2, B, L, s));
    AID_TRY(ln_bwd_rows(dh1, L2, sv.s0.lin, L2, sv.s0.stats, e.out1.g, e.out1.b, ACT_MISH, train ? drop_out : nullptr, d_o0, L2, dyeff, B, L2, s));
    AID_TRY(ln_param_grad(dyeff, sv.s0.lin, L2, sv.s0.stats, gr.out1.g, gr.out1.b, B, L2, s, &ws));
    // ---- Linear(F -> 2L): dW [2L, F] += d_o0^T xln ; db ; dxln = d_o0 W
    AID_TRY(linear_bwd_weight(d_o0, L2, sv.xln, (int)g.F, gr.out0.w, B, L2, (int)g.F, s, 1));
    AID_TRY(colsum(d_o0, L2, gr.out0.b, B, L2, 1, s, &ws));
    AID_TRY(linear_bwd_data(d_o0, L2, e.out0.w, (int)g.F, dxln, (int)g.F, B, L2, s));
    // ---- LayerNorm(F)
    hipLaunchKernelGGL(bigln_param_grad_kernel, dim3((unsigned)((g.F + 255) / 256)), dim3(256), 0, s, dxln, sv.act[3], sv.att, HW, sv.ln_stats, gr.ln_g, gr.ln_b, B, g.F);
    {
        const int vec = HW % 4 == 0 && ((((uintptr_t)dxln | (uintptr_t)sv.act[3] | (uintptr_t)sv.att | (uintptr_t)e.ln_g | (uintptr_t)dxf) & 15) == 0);
        hipLaunchKernelGGL(bigln_bwd_kernel, dim3(B), dim3(1024), 0, s, dxln, sv.act[3], sv.att, B, HW, e.ln_g, sv.ln_stats, dxf, g.F, vec);
    }
    AID_LAUNCH_CHECK();
    }
    if (phase == 1) return AID_OK;
    // ---- spatial attention
    const size_t nact3 = (size_t)256 * B * HW;
    float* dact = ws.f32(nact3); float* da = ws.f32((size_t)B * HW); float* tpart = ws.f32((size_t)B * HW);
    AID_WS_OK(ws);
    L1D(attn_bwd_a_kernel, B * HW, s, dxf, sv.act[3], sv.att, sv.apre, e.att_temperature, dact, da, tpart, 256, B, HW);
    float* cpart = ws.f32((size_t)ATTN_C_PARTS * 100); AID_WS_OK(ws);
    hipLaunchKernelGGL(attn_bwd_c_part_kernel, dim3(ATTN_C_PARTS), dim3(256), 0, s, da, sv.pool, tpart, cpart, B, Hh, Hh);
    hipLaunchKernelGGL(attn_bwd_c_fold_kernel, dim3(100), dim3(64), 0, s, cpart, ATTN_C_PARTS, gr.att_conv_w, gr.att_conv_b, gr.att_temperature);
    AID_LAUNCH_CHECK();
    if (HW % 4 == 0 && (((uintptr_t)dact & 15) == 0)) {
        const size_t nq = (size_t)B * HW / 4;
        hipLaunchKernelGGL(attn_bwd_b4_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(256), 0, s, da, e.att_conv_w, sv.amax, dact, 256, B, Hh, Hh);
    } else L1D(attn_bwd_b_kernel, B * HW, s, da, e.att_conv_w, sv.amax, dact, 256, B, Hh, Hh);
    // ---- conv stack, reverse
    float* dcur = dact;      // gradient wrt act[i] (CNHW)
    for (int i = ENC_LAYERS - 1; i >= 0; --i) {
        const int hw = g.ho[i] * g.ho[i];
        const size_t N = (size_t)B * hw;
        const size_t mark = ws.mark();
        float* dconv = ws.f32((size_t)g.cout[i] * N);
        const int ldwt = (g.cout[i] * 9 + 3) / 4 * 4;
        float* wflip = i > 0 ? ws.f32((size_t)g.cin[i] * ldwt) : nullptr;
        float* dweff = ws.f32((size_t)g.cout[i] * g.Kp[i]);
        const int S = conv_dw_slabs(N, g.cout[i], g.Kp[i]);
        float* slab = ws.f32((size_t)S * g.cout[i] * g.Kp[i]);
        float* ppart = ws.f32((size_t)B * g.cout[i] * 2);
        float* snpart = ws.f32(SN_BWD_PARTS);
        AID_WS_OK(ws);
        if (gn_reg_ok(g.cout[i], g.G[i], hw, sv.conv_out[i], dcur) && ((((uintptr_t)dconv) & 15) == 0)) {
            if (g.cout[i] / g.G[i] == 4) hipLaunchKernelGGL(gn_mish_bwd_reg_kernel<4>, dim3(B * g.G[i]), dim3(256), 0, s, sv.conv_out[i], e.gn_g[i], e.gn_b[i], masks[i], sv.gn_stats[i], dcur, dconv, ppart, g.cout[i], B, hw, g.G[i]);
            else hipLaunchKernelGGL(gn_mish_bwd_reg_kernel<8>, dim3(B * g.G[i]), dim3(256), 0, s, sv.conv_out[i], e.gn_g[i], e.gn_b[i], masks[i], sv.gn_stats[i], dcur, dconv, ppart, g.cout[i], B, hw, g.G[i]);
            hipLaunchKernelGGL(gn_param_fold_kernel, dim3((g.cout[i] + 63) / 64), dim3(64), 0, s, ppart, gr.gn_g[i], gr.gn_b[i], g.cout[i], B);
        } else {
            hipLaunchKernelGGL(gn_mish_bwd_kernel, dim3(B * g.G[i]), dim3(256), 0, s, sv.conv_out[i], e.gn_g[i], e.gn_b[i], masks[i], sv.gn_stats[i], dcur, dconv,
                               g.cout[i], B, hw, g.G[i]);
            hipLaunchKernelGGL(gn_param_grad_kernel, dim3(g.cout[i]), dim3(1024), 0, s, dcur, sv.conv_out[i], sv.gn_stats[i], gr.gn_g[i], gr.gn_b[i], g.cout[i], B, hw, g.G[i]);
        }
        AID_LAUNCH_CHECK();
        // dW_eff[Cout][Kp] = sum_n dconv[Cout][n] X(k, n): the layer's input is gathered again (implicit GEMM; huge reduction -> slabs)
        AID_TRY(conv3_igemm_dw(dconv, i > 0 ? sv.act[i - 1] : sv.x0, dweff, g.Kp[i], g.cin[i], g.cout[i], B, g.hi[i], g.ho[i], g.stride[i], slab, S, s));
        hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(SN_BWD_PARTS), dim3(256), 0, s, dweff, sv.w_eff[i], g.cout[i], g.K[i], g.Kp[i], snpart);
        hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3((g.cout[i] * g.K[i] + 255) / 256), dim3(256), 0, s, dweff, snpart, sv.sn[i], g.cout[i], g.K[i], g.Kp[i], gr.conv_w[i]);
        AID_LAUNCH_CHECK();
        if (i > 0) {   // gradient wrt act[i-1]: the stride-1 conv of dconv with the flipped, channel-transposed weights
            AID_REQUIRE(g.stride[i] == 1, AID_E_SHAPE, "aid_encoder_backward: data gradient of a strided conv is not built (layer %d)", i);
            const int tmj = conv3_tapmajor_ok(g.cout[i], g.cin[i]) ? 1 : 0;      // the data-gradient conv contracts over (tap, co)
            AID_TRY(conv3_flip_weights(sv.w_eff[i], g.Kp[i], wflip, ldwt, g.cin[i], g.cout[i], s, tmj));
            // dact (gradient wrt act[i]) was consumed by the two GroupNorm kernels above: the gradient wrt act[i-1] goes straight into it
            // (cin*hi*hi*B <= 256*HW*B for layers 1..3)
            AID_TRY(conv3_igemm_ex(wflip, ldwt, dconv, dact, g.cout[i], g.cin[i], B, g.ho[i], g.hi[i], 1, tmj, s));
            dcur = dact;
        }
        ws.release(mark);
    }
    return AID_OK;
}

int aid_encoder_backward(const AidEncoderNet* enc, const AidEncoderNet* grads, int B, const float* d_features, const float* drop2d_1,
                         const float* drop2d_2, const float* drop_out, int train, const void* saved, size_t saved_bytes, void* wsp,
                         size_t ws_bytes, void* stream) {
    return encoder_backward_impl(enc, grads, B, d_features, drop2d_1, drop2d_2, drop_out, train, saved, saved_bytes, wsp, ws_bytes, 0, stream);
}
int aid_encoder_backward_phase(const AidEncoderNet* enc, const AidEncoderNet* grads, int B, const float* d_features, const float* drop2d_1,
                               const float* drop2d_2, const float* drop_out, int train, const void* saved, size_t saved_bytes, void* wsp,
                               size_t ws_bytes, int phase, void* stream) {
    return encoder_backward_impl(enc, grads, B, d_features, drop2d_1, drop2d_2, drop_out, train, saved, saved_bytes, wsp, ws_bytes, phase, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// pixel-mode training pieces: feature decoder reconstruction, InfoNCE, stand-alone random shift
namespace aid {
__global__ void random_shift_nchw_kernel(const float* __restrict__ x, const int32_t* __restrict__ sh, const int32_t* __restrict__ sw, int pad,
                                         float* __restrict__ out, int B, int C, int H, int W) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * C * H * W) return;
    const int w = i % W, h = (i / W) % H; const size_t bc = i / ((size_t)W * H); const int b = bc / C;
    out[i] = x[(bc * H + reflect_idx(h + sh[b] - pad, H)) * W + reflect_idx(w + sw[b] - pad, W)];
}
// pred = tanh(lin); loss = mean (pred - t)^2 ; dlin = 2 (pred - t)/(n) * (1 - pred^2) ; d_target += -2 (pred - t)/n
__global__ void fd_loss_kernel(const float* __restrict__ lin, const float* __restrict__ t, float* __restrict__ dlin, float* __restrict__ dt,
                               float* __restrict__ loss, int n) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float p = tanhf(lin[i]), d = p - t[i];
        acc += d * d;
        const float g = 2.f * d / (float)n;
        dlin[i] = g * (1.f - p * p);
        if (dt) dt[i] -= g;
    }
    const float s = wg_sum256(acc, sh);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}
// rows normalised: out = x / max(||x||, 1e-12) ; norms saved
__global__ void row_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ nrm, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += x[(size_t)b * L + l] * x[(size_t)b * L + l];
    const float n = fmaxf(sqrtf(s), 1e-12f);
    nrm[b] = n;
    for (int l = 0; l < L; ++l) out[(size_t)b * L + l] = x[(size_t)b * L + l] / n;
}
// cross entropy over rows of logits [B,B] against the diagonal; dlogits = (softmax - onehot)/B
__global__ void ce_diag_kernel(const float* __restrict__ logits, float* __restrict__ dlogits, float* __restrict__ loss, int B) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int r = threadIdx.x; r < B; r += blockDim.x) {
        const float* lr = logits + (size_t)r * B;
        float mx = lr[0];
        for (int c = 1; c < B; ++c) mx = fmaxf(mx, lr[c]);
        float se = 0.f;
        for (int c = 0; c < B; ++c) se += expf(lr[c] - mx);
        const float lse = mx + logf(se);
        acc += lse - lr[r];
        for (int c = 0; c < B; ++c) dlogits[(size_t)r * B + c] = (expf(lr[c] - lse) - (c == r ? 1.f : 0.f)) / (float)B;
    }
    const float s = wg_sum256(acc, sh);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}
// d(x) of y = x/||x||:  (dy - y (dy.y)) / ||x|| ; accumulate weight * that
__global__ void normalize_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ nrm, float weight,
                                     float* __restrict__ dx, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float dot = 0.f;
    for (int l = 0; l < L; ++l) dot += dy[(size_t)b * L + l] * y[(size_t)b * L + l];
    for (int l = 0; l < L; ++l) dx[(size_t)b * L + l] += weight * (dy[(size_t)b * L + l] - y[(size_t)b * L + l] * dot) / nrm[b];
}
}  // namespace aid

extern "C" {

int aid_random_shift(const float* x, const int32_t* shift_h, const int32_t* shift_w, int pad, int B, int C, int H, int W, float* out, void* stream) {
    AID_REQUIRE(x && shift_h && shift_w && out, AID_E_NULL, "aid_random_shift: null tensor");
    hipStream_t s = (hipStream_t)stream;
    L1D(random_shift_nchw_kernel, (size_t)B * C * H * W, s, x, shift_h, shift_w, pad, out, B, C, H, W);
    return AID_OK;
}

int aid_feature_decoder_step(const AidFeatureDecoder* fd, const AidFeatureDecoder* fgrads, int B, const float* latents, const float* target,
                             const float* drop_mask, float* d_target, float* loss_out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(fd && fgrads && latents && target && loss_out, AID_E_NULL, "aid_feature_decoder_step: null argument");
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const int L = fd->l0.in, D = fd->l0.out, Lo = fd->l4.out;
    LlaSave s0; lla_alloc(B, D, ws, s0);
    float* lin = ws.f32((size_t)B * Lo); float* dlin = ws.f32((size_t)B * Lo); float* dh = ws.f32((size_t)B * D);
    AID_WS_OK(ws);
    AID_TRY(lla_fwd(fd->l0, fd->ln, ACT_RELU, drop_mask, latents, L, B, s0, s));
    AID_TRY(linear_fwd(s0.y, D, fd->l4.w, fd->l4.b, lin, Lo, B, Lo, D, ACT_NONE, s));
    hipLaunchKernelGGL(fd_loss_kernel, dim3(1), dim3(1024), 0, s, lin, target, dlin, d_target, loss_out, B * Lo);
    AID_LAUNCH_CHECK();
    AID_TRY(linear_bwd_weight(dlin, Lo, s0.y, D, fgrads->l4.w, B, Lo, D, s, 1));
    AID_TRY(colsum(dlin, Lo, fgrads->l4.b, B, Lo, 1, s, &ws));
    AID_TRY(linear_bwd_data(dlin, Lo, fd->l4.w, D, dh, D, B, Lo, s));
    AID_TRY(lla_bwd(fd->l0, fd->ln, &fgrads->l0, &fgrads->ln, ACT_RELU, drop_mask, latents, L, B, s0, dh, nullptr, 0, ws, s));
    return AID_OK;
}

int aid_feature_decoder_forward(const AidFeatureDecoder* fd, int B, const float* latents, const float* drop_mask, float* out, void* wsp, size_t ws_bytes,
                                void* stream) {
    AID_REQUIRE(fd && latents && out, AID_E_NULL, "aid_feature_decoder_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const int L = fd->l0.in, D = fd->l0.out, Lo = fd->l4.out;
    LlaSave s0; lla_alloc(B, D, ws, s0); AID_WS_OK(ws);
    AID_TRY(lla_fwd(fd->l0, fd->ln, ACT_RELU, drop_mask, latents, L, B, s0, s));
    return linear_fwd(s0.y, D, fd->l4.w, fd->l4.b, out, Lo, B, Lo, D, ACT_TANH, s);
}

int aid_infonce_step(const AidDynamicsNet* dyn, int B, const float* latents, const float* actions, const float* next_enc, float weight,
                     float* d_next_enc, float* loss_out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dyn && latents && actions && next_enc && d_next_enc && loss_out, AID_E_NULL, "aid_infonce_step: null argument");
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const int L = dyn->L;
    float* pred = ws.f32((size_t)B * L); float* pn = ws.f32((size_t)B * L); float* tn = ws.f32((size_t)B * L); float* n1 = ws.f32(B); float* n2 = ws.f32(B);
    float* logits = ws.f32((size_t)B * B); float* dlog = ws.f32((size_t)B * B); float* dtn = ws.f32((size_t)B * L);
    DynSave sv;
    AID_TRY(dynamics_fwd(*dyn, B, latents, L, actions, dyn->A, pred, sv, ws, s));
    L1D(row_normalize_kernel, B, s, pred, pn, n1, B, L);
    L1D(row_normalize_kernel, B, s, next_enc, tn, n2, B, L);
    const float scale = (1.0f / 0.1f) * (1.0f / (1.0f + expf(0.5f * logf(0.1f))));     // 1/temperature * 1/(1 + mean std), std = sqrt(0.1)
    {   // logits = pn tn^T * scale
        GemmArgs g = gemm_args(pn, L, tn, L, logits, B, B, B, L); g.alpha = scale;
        AID_TRY(gemm_f32(g, s));
    }
    hipLaunchKernelGGL(ce_diag_kernel, dim3(1), dim3(1024), 0, s, logits, dlog, loss_out, B);
    AID_LAUNCH_CHECK();
    {   // dtn[j,:] = scale * sum_i dlog[i,j] pn[i,:]   (A = dlog stored [K=i][M=j] -> k-major; B = pn stored [K=i][N=l] -> k-major)
        GemmArgs g = gemm_args(dlog, B, pn, L, dtn, L, B, L, B); g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = scale;
        AID_TRY(gemm_f32(g, s));
    }
    L1D(normalize_bwd_kernel, B, s, dtn, tn, n2, weight, d_next_enc, B, L);
    return AID_OK;
}


size_t aid_conv_decoder_workspace(const AidConvDecoder* d, int R) {
    if (!d || R <= 0) return 0;
    const size_t H = d->hidden, hw = 21 * 21;
    size_t n = 0;
    n += (size_t)R * (4 * H) * 2 + (size_t)R * H * hw * 2;                  // latent_proj activations
    {   // W_eff images of every conv (conv_decoder_fwd's prep_weight calls: [cout][9 cin padded to 32] each, + a [cout] vector)
        const size_t cin[4] = {H, H, H / 2, H / 4}, cout[4] = {H, H / 2, H / 4, H / 8};
        auto img = [](size_t ci, size_t co, size_t k) { return co * ((ci * k * k + 31) / 32 * 32) + co + 128; };
        for (int i = 0; i < 4; ++i) n += img(cin[i], i ? 4 * cout[i] : cout[i], 3) + img(cout[i], cout[i], 3) + (i ? img(cin[i], 4 * cout[i], 1) : 0);
        n += img(H / 8, 32, 3) + img(32, 32, 3) + img(32, 3, 3) + 4096;
    }
    const size_t amax = (size_t)R * (3528 * H > 32u * 168 * 168 ? 3528 * H : 32u * 168 * 168);   // largest stage: (H/8) x 168^2 = 8H x 21^2 per image
    n += amax * 5;                                                          // ping-pong activations + conv outputs
    return sizeof(float) * n + (1 << 20);
}

int aid_conv_decoder_forward(const AidConvDecoder* dec, int R, const float* z, float* out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dec && z && out, AID_E_NULL, "aid_conv_decoder_forward: null argument");
    AID_REQUIRE(R > 0 && R <= 64 && dec->hidden >= 64 && dec->hidden % 32 == 0 && dec->L > 0, AID_E_SHAPE,
                "aid_conv_decoder_forward: R=%d (chunk rows to <= 64), hidden=%d", R, dec->hidden);
    Workspace ws(wsp, ws_bytes);
    return conv_decoder_fwd(*dec, R, z, out, ws, (hipStream_t)stream);
}
}  // extern "C"

namespace aid {
// a[r][c][p] <-> b[c][r][p]  (NCHW <-> the conv stack's CNHW); to_cnhw: out = CNHW image of an NCHW input, else the reverse
__global__ void nchw_cnhw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int R, int HW, int to_cnhw) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * R * HW) return;
    const int p = i % HW, r = (i / HW) % R, c = i / ((size_t)HW * R);          // i indexes the CNHW image
    const size_t j = ((size_t)r * C + c) * HW + p;
    if (to_cnhw) out[i] = in[j]; else out[j] = in[i];
}
}  // namespace aid

extern "C" {
// DecoderBlock.forward, eval mode (encoder/visual_encoders.py:370-459): main = IN(conv3x3_sn(Mish(IN(PixelShuffle?(conv3x3_sn(x)))))) (Dropout2d: identity in
// eval), residual = identity | IN(conv1x1_sn(x)) | IN(PixelShuffle(conv1x1_sn(x))), out = Mish(main + residual).  x [R][cin][h][h] -> out [R][cout][ho][ho],
// ho = 2 h when upsampling.  The spectral-norm weights are W / (u . W v) of the stored u, v (no power iteration in eval mode).
size_t aid_decoder_block_workspace(int cin, int cout, int upsample, int R, int h) {
    if (cin <= 0 || cout <= 0 || R <= 0 || h <= 0) return 0;
    const size_t ho = upsample ? 2 * (size_t)h : h, c0 = upsample ? 4 * (size_t)cout : cout;
    auto img = [](size_t ci, size_t co, size_t k) { return co * ((ci * k * k + 31) / 32 * 32) + co + 128; };
    size_t n = img(cin, c0, 3) + img(cout, cout, 3) + img(cin, c0, 1);
    n += (size_t)R * ((size_t)cin * h * h + 5 * (size_t)cout * ho * ho + 2 * c0 * h * h);
    return sizeof(float) * n + (1 << 20);
}
int aid_decoder_block_forward(const AidDecoderBlock* blk, int cin, int cout, int upsample, int R, int h, const float* x, float* out, void* wsp, size_t ws_bytes,
                              void* stream) {
    AID_REQUIRE(blk && x && out && blk->main0.w && blk->main1.w, AID_E_NULL, "aid_decoder_block_forward: null argument");
    AID_REQUIRE(cin > 0 && cout > 0 && R > 0 && h >= 4 && h <= 168 && (size_t)R * cin * h * h < ((size_t)1 << 30) && (size_t)R * cout * 4 * h * h < ((size_t)1 << 30),
                AID_E_SHAPE, "aid_decoder_block_forward: cin=%d cout=%d R=%d h=%d", cin, cout, R, h);
    AID_REQUIRE(blk->res.w || (cin == cout && !upsample), AID_E_SHAPE, "aid_decoder_block_forward: a block that changes channels or upsamples needs its residual conv");
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const int up = upsample ? 1 : 0, ho = up ? 2 * h : h, c0 = up ? 4 * cout : cout;
    ConvW m0, m1, rs;
    AID_TRY(prep_weight(blk->main0, cin, c0, 3, ws, m0, s));
    AID_TRY(prep_weight(blk->main1, cout, cout, 3, ws, m1, s));
    if (blk->res.w) AID_TRY(prep_weight(blk->res, cin, c0, 1, ws, rs, s));
    const size_t n_in = (size_t)R * cin * h * h, n_out = (size_t)R * cout * ho * ho, n_c0 = (size_t)R * c0 * h * h;
    float* xc = ws.f32(n_in); float* cv = ws.f32(n_c0 > n_out ? n_c0 : n_out); float* y = ws.f32(n_out); float* cr = ws.f32(n_c0); float* rn = ws.f32(n_out); float* yc = ws.f32(n_out);
    AID_WS_OK(ws);
    L1D(nchw_cnhw_kernel, n_in, s, x, xc, cin, R, h * h, 1);
    AID_TRY(conv3_fwd(m0, xc, R, h, cv, s));
    AID_TRY(instnorm(cv, blk->main0.bias, (const float*)nullptr, y, cout, R, ho, ho, up, ACT_MISH, s));
    AID_TRY(conv3_fwd(m1, y, R, ho, cv, s));
    const float* resid = xc;
    if (blk->res.w) {
        AID_TRY(conv1_fwd(rs, xc, R, h, cr, s));
        AID_TRY(instnorm(cr, blk->res.bias, (const float*)nullptr, rn, cout, R, ho, ho, up, ACT_NONE, s));
        resid = rn;
    }
    AID_TRY(instnorm(cv, blk->main1.bias, resid, yc, cout, R, ho, ho, 0, ACT_MISH, s));
    L1D(nchw_cnhw_kernel, n_out, s, yc, out, cout, R, ho * ho, 0);
    return AID_OK;
}
}  // extern "C"

namespace aid {
// The decoder in three parts, so that a caller decoding many latents (the pixel estimator: z and four perturbed copies of every row, 16 images at a
// time) prepares the spectral-norm weight images ONCE and streams latent_proj's 1.85 GB second Linear ONCE per <= 640 rows as an MFMA GEMM, instead of
// once per 16-image decode through the weight-streaming skinny kernel (40 x 0.64 ms per 128 rows).
struct DecPrep { ConvW m0[4], m1[4], rs[4], o0, o3, o6; };
static int conv_decoder_prep(const AidConvDecoder& d, Workspace& ws, DecPrep& p, hipStream_t s) {
    const int H = d.hidden;
    // ---- conv weights (eval-mode spectral norm: W / (u . W v))
    const int cin[4] = {H, H, H / 2, H / 4}, cout[4] = {H, H / 2, H / 4, H / 8}; const int up[4] = {0, 1, 1, 1};
    for (int i = 0; i < 4; ++i) {
        AID_TRY(prep_weight(d.blk[i].main0, cin[i], up[i] ? 4 * cout[i] : cout[i], 3, ws, p.m0[i], s));
        AID_TRY(prep_weight(d.blk[i].main1, cout[i], cout[i], 3, ws, p.m1[i], s));
        if (up[i]) AID_TRY(prep_weight(d.blk[i].res, cin[i], 4 * cout[i], 1, ws, p.rs[i], s));
    }
    AidSNConv t0 = {d.out0_w, nullptr, nullptr, d.out0_b}, t3 = {d.out3_w, nullptr, nullptr, d.out3_b}, t6 = {d.out6_w, nullptr, nullptr, d.out6_b};
    AID_TRY(prep_weight(t0, H / 8, 32, 3, ws, p.o0, s)); AID_TRY(prep_weight(t3, 32, 32, 3, ws, p.o3, s)); AID_TRY(prep_weight(t6, 32, 3, 3, ws, p.o6, s));
    return AID_OK;
}
// latent_proj up to the second Linear: Linear, LN, Mish, (Dropout: eval), Linear -> big [R][H * 441] (the LayerNorm over 112 896 + Mish follow per chunk)
static int conv_decoder_latent(const AidConvDecoder& d, int R, const float* z, float* big, Workspace& ws, hipStream_t s) {
    const int H = d.hidden, L = d.L, HW0 = 441;
    const size_t mark = ws.mark();
    float* a0 = ws.f32((size_t)R * 4 * H); float* a1 = ws.f32((size_t)R * 4 * H);
    AID_WS_OK(ws);
    AID_TRY(linear_fwd(z, L, d.l0.w, d.l0.b, a0, 4 * H, R, 4 * H, L, ACT_NONE, s));
    AID_TRY(ln_act(a0, 4 * H, d.ln1.g, d.ln1.b, ACT_MISH, nullptr, a1, 4 * H, nullptr, R, 4 * H, s));
    AID_TRY(linear_fwd(a1, 4 * H, d.l4.w, d.l4.b, big, H * HW0, R, H * HW0, 4 * H, ACT_NONE, s));
    ws.release(mark);
    return AID_OK;
}
static int conv_decoder_trunk(const AidConvDecoder& d, const DecPrep& p, int R, const float* big, float* out, Workspace& ws, hipStream_t s) {
    const size_t mark0 = ws.mark();
    const int H = d.hidden, HW0 = 441;
    const int cout[4] = {H, H / 2, H / 4, H / 8}; const int up[4] = {0, 1, 1, 1};
    const ConvW* m0 = p.m0; const ConvW* m1 = p.m1; const ConvW* rs = p.rs; const ConvW& o0 = p.o0; const ConvW& o3 = p.o3; const ConvW& o6 = p.o6;
    const size_t amax = (size_t)R * ((size_t)3528 * H > 32u * 168 * 168 ? (size_t)3528 * H : 32u * 168 * 168);
    float* xa = ws.f32(amax); float* xb = ws.f32(amax); float* cv = ws.f32(amax); float* cr = ws.f32(2 * amax);   // cr: 1x1-conv output + its IN
    AID_WS_OK(ws);
    AID_TRY(bigln_mish_cnhw(big, d.ln5.g, d.ln5.b, xa, R, H, HW0, ws, s));
    AID_LAUNCH_CHECK();
    // ---- decoder blocks
    int hsz = 21;
    float* x = xa; float* y = xb;
    for (int i = 0; i < 4; ++i) {
        const int ho = up[i] ? 2 * hsz : hsz, co = cout[i];
        AID_TRY(conv3_fwd(m0[i], x, R, hsz, cv, s));                                                       // [co or 4co][R][hsz^2]
        AID_TRY(instnorm(cv, d.blk[i].main0.bias, (const float*)nullptr, y, co, R, ho, ho, up[i], ACT_MISH, s));
        AID_TRY(conv3_fwd(m1[i], y, R, ho, cv, s));                                                        // [co][R][ho^2]
        const float* resid = x;                                                                                 // identity residual (block 0)
        if (up[i]) {
            AID_TRY(conv1_fwd(rs[i], x, R, hsz, cr, s));                                                        // [4co][R][hsz^2]
            AID_TRY(instnorm(cr, d.blk[i].res.bias, (const float*)nullptr, cr + (size_t)4 * co * R * hsz * hsz,
                               co, R, ho, ho, 1, ACT_NONE, s));
            resid = cr + (size_t)4 * co * R * hsz * hsz;
        }
        // main = IN(conv2 + bias); out = Mish(main + residual)
        AID_TRY(instnorm(cv, d.blk[i].main1.bias, resid, y, co, R, ho, ho, 0, ACT_MISH, s));
        float* t = x; x = y; y = t; hsz = ho;
    }
    // ---- output_proj: (conv, IN, Mish) x2, conv, sigmoid
    AID_TRY(conv3_fwd(o0, x, R, hsz, cv, s));
    AID_TRY(instnorm(cv, d.out0_b, (const float*)nullptr, y, 32, R, hsz, hsz, 0, ACT_MISH, s));
    AID_TRY(conv3_fwd(o3, y, R, hsz, cv, s));
    AID_TRY(instnorm(cv, d.out3_b, (const float*)nullptr, x, 32, R, hsz, hsz, 0, ACT_MISH, s));
    AID_TRY(conv3_fwd(o6, x, R, hsz, cv, s));
    L1D(sigmoid_nchw_kernel, (size_t)3 * R * hsz * hsz, s, cv, d.out6_b, out, 3, R, hsz * hsz);
    ws.release(mark0);
    return AID_OK;
}
int conv_decoder_fwd(const AidConvDecoder& d, int R, const float* z, float* out, Workspace& ws, hipStream_t s) {
    const size_t mark0 = ws.mark();
    DecPrep p;
    AID_TRY(conv_decoder_prep(d, ws, p, s));
    float* big = ws.f32((size_t)R * d.hidden * 441); AID_WS_OK(ws);
    AID_TRY(conv_decoder_latent(d, R, z, big, ws, s));
    AID_TRY(conv_decoder_trunk(d, p, R, big, out, ws, s));
    ws.release(mark0);
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Pixel branch of the epistemic estimator, feature side (active_inference.py:865-883,940-993; utils/util.py:7-75):
// (decoder(z + delta) - decoder(z)) / eps  ->  three 5x5 stride-2 convs + ReLU  ->  SpatialAttentionAggregator  ->  jin [R, 4*256]
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void perturb_rows_kernel(const float* __restrict__ z, const float* __restrict__ d, const float* __restrict__ scale, float* __restrict__ zp, int R, int L) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float n2 = 0.f;
    for (int l = 0; l < L; ++l) { const float v = d[(size_t)r * L + l]; n2 += v * v; }
    const float inv = scale[0] / fmaxf(sqrtf(n2), 1e-12f);
    for (int l = 0; l < L; ++l) zp[(size_t)r * L + l] = z[(size_t)r * L + l] + d[(size_t)r * L + l] * inv;
}
// diff[c][r][p] = (fp[r][c][p] - fz[r][c][p]) / eps        (NCHW images -> CNHW difference)
__global__ void fd_to_cnhw_kernel(const float* __restrict__ fp, const float* __restrict__ fz, const float* __restrict__ eps, float* __restrict__ out, int C, int R, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * R * HW) return;
    const int p = i % HW, r = (i / HW) % R, c = i / ((size_t)HW * R);
    const size_t src = ((size_t)r * C + c) * HW + p;
    out[i] = (fp[src] - fz[src]) / eps[0];
}
// im2col for 5x5, stride 2, pad 2; x CNHW [Cin][R][Hi][Hi]; col [Kp][R*Ho*Ho], k = ci*25 + kh*5 + kw
__global__ void im2col5s2_kernel(const float* __restrict__ x, float* __restrict__ col, int Cin, int R, int Hi, int Ho, int K, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t N = (size_t)R * Ho * Ho;
    const int k = i / N; const size_t n = i % N;
    float v = 0.f;
    if (k < K) {
        const int ci = k / 25, kh = (k % 25) / 5, kw = k % 5;
        const int wo = n % Ho, ho = (n / Ho) % Ho, r = n / ((size_t)Ho * Ho);
        const int hi = 2 * ho + kh - 2, wi = 2 * wo + kw - 2;
        if (hi >= 0 && hi < Hi && wi >= 0 && wi < Hi) v = x[(((size_t)ci * R + r) * Hi + hi) * Hi + wi];
    }
    col[i] = v;
}
// x[c][n] = act(x[c][n] + bias[c])
__global__ void bias_rows_act_kernel(float* __restrict__ x, const float* __restrict__ bias, int C, size_t N, int act) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * N) return;
    x[i] = act_apply(x[i] + bias[i / N], act);
}
// tokens: x[c][r][t] += pos[t][c]
__global__ void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, int C, int R, int T) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * R * T) return;
    const int t = i % T, c = i / ((size_t)T * R);
    x[i] += pos[(size_t)t * C + c];
}
// nn.MultiheadAttention with 16 query tokens over T key tokens, one workgroup per (row r, head h), head dim 16:
// kv CNHW-like [2*C][R][T] (K rows 0..C-1, V rows C..2C-1, biases already added); q [16][C] (shared by every row, bias added);
// mask [R*heads][16][T] scaled keep mask or NULL; out [R][16][C]
__global__ void __launch_bounds__(256) agg_attention_kernel(const float* __restrict__ kv, const float* __restrict__ q, const float* __restrict__ mask,
                                                             float* __restrict__ out, int C, int R, int T, int heads) {
    extern __shared__ float sm[];
    constexpr int NQ = 16, HD = 16;
    float* sc = sm;                       // [NQ][T]
    float* qs = sm + (size_t)NQ * T;      // [NQ][HD]
    float* red = qs + NQ * HD;            // [NQ][2]
    const int r = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    for (int i = tid; i < NQ * HD; i += blockDim.x) qs[i] = q[(size_t)(i / HD) * C + h * HD + i % HD] * 0.25f;     // 1/sqrt(head_dim), applied to q like torch
    __syncthreads();
    const float* Kh = kv + ((size_t)(h * HD) * R + r) * T;           // row d at + d * R * T
    const float* Vh = kv + ((size_t)(C + h * HD) * R + r) * T;
    for (int i = tid; i < NQ * T; i += blockDim.x) {
        const int qi = i / T, t = i % T;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(qs[qi * HD + d], Kh[(size_t)d * R * T + t], a);
        sc[i] = a;
    }
    __syncthreads();
    // softmax over T per query: one wave per 4 queries
    const int lane = tid & 63, wave = tid >> 6;
    for (int qi = wave; qi < NQ; qi += 4) {
        float mx = -3.4e38f;
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, sc[qi * T + t]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int t = lane; t < T; t += 64) { const float e = expf(sc[qi * T + t] - mx); sc[qi * T + t] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        const float* mk = mask ? mask + ((size_t)(r * heads + h) * NQ + qi) * T : nullptr;
        for (int t = lane; t < T; t += 64) sc[qi * T + t] = sc[qi * T + t] * inv * (mk ? mk[t] : 1.0f);
    }
    __syncthreads();
    // out[qi][d] = sum_t P[qi][t] V[d][t] : thread (qi, d)
    {
        const int qi = tid / HD, d = tid % HD;
        float a = 0.f;
        const float* v = Vh + (size_t)d * R * T;
        for (int t = 0; t < T; ++t) a = fmaf(sc[qi * T + t], v[t], a);
        out[((size_t)r * NQ + qi) * C + h * HD + d] = a;
    }
    (void)red;
}

constexpr int JAC_SC = 128;      // rows whose latent_proj runs as one GEMM in pixel_jacobian_features (5 x 128 x 225 792 floats = 578 MB at hidden = 512)
struct PlainConv { const float* w_eff; int cin, cout, K, Kp; };
static int prep_plain(const float* w, int cin, int cout, int k, Workspace& ws, PlainConv& out, hipStream_t s) {
    const int K = cin * k * k, Kp = (K + 31) / 32 * 32;
    float* w_eff = ws.f32((size_t)cout * Kp); AID_WS_OK(ws);
    L1D(pad_weight_kernel, (size_t)cout * Kp, s, w, cout, K, Kp, w_eff, 0);
    out.w_eff = w_eff; out.cin = cin; out.cout = cout; out.K = K; out.Kp = Kp;
    return AID_OK;
}

int pixel_jacobian_features(const AidConvDecoder& dec, const AidPixelEpistemicNet& e, int R, const float* z_all, const float* delta,
                            const float* attn_mask, int train, uint64_t seed, uint64_t offset, float* jin, Workspace& ws, hipStream_t s) {
    constexpr int RC = 16, C = 128, T = 441, NQ = 16, HEADS = 8;
    static_assert(JAC_SC % RC == 0, "super-chunks are whole chunks");
    const int L = dec.L;
    const size_t mark = ws.mark();
    PlainConv p0, p2, p4;
    AID_TRY(prep_plain(e.pp0_w, 3, 32, 5, ws, p0, s)); AID_TRY(prep_plain(e.pp2_w, 32, 64, 5, ws, p2, s)); AID_TRY(prep_plain(e.pp4_w, 64, 128, 5, ws, p4, s));
    // q = queries Wq^T + bq  (identical for every row)
    float* q = ws.f32((size_t)NQ * C); AID_WS_OK(ws);
    AID_TRY(linear_fwd(e.queries, C, e.in_proj_w, e.in_proj_b, q, C, NQ, C, C, ACT_NONE, s));
    float* zp = ws.f32((size_t)RC * L); float* fz = ws.f32((size_t)RC * 3 * 168 * 168); float* fp = ws.f32((size_t)RC * 3 * 168 * 168);
    float* dimg = ws.f32((size_t)3 * RC * 168 * 168);
    float* c1 = ws.f32((size_t)32 * RC * 84 * 84); float* c2 = ws.f32((size_t)64 * RC * 42 * 42); float* c3 = ws.f32((size_t)C * RC * T);
    float* col = ws.f32((size_t)p2.Kp * RC * 42 * 42 > (size_t)p0.Kp * RC * 84 * 84 ? (size_t)p2.Kp * RC * 42 * 42 : (size_t)p0.Kp * RC * 84 * 84);
    float* kv = ws.f32((size_t)2 * C * RC * T); float* att = ws.f32((size_t)RC * NQ * C); float* ao = ws.f32((size_t)RC * NQ * C);
    float* a0 = ws.f32((size_t)RC * e.agg0.out); float* a1 = ws.f32((size_t)RC * e.agg0.out);
    float* mask_buf = (train && !attn_mask) ? ws.f32((size_t)RC * HEADS * NQ * T) : nullptr;
    AID_WS_OK(ws);
    uint64_t ctr = offset;
    // decoder weights once; latent_proj for z and its four perturbed copies of up to SC rows at a time (5 SC x 225 792 floats), then 16 images per trunk pass
    DecPrep prep;
    AID_TRY(conv_decoder_prep(dec, ws, prep, s));
    const size_t bigw = (size_t)dec.hidden * 441;
    float* zall = ws.f32((size_t)5 * JAC_SC * L); float* big = ws.f32((size_t)5 * JAC_SC * bigw);
    AID_WS_OK(ws);
    (void)zp;
    for (int r0 = 0; r0 < R; r0 += RC) {
        const int n = R - r0 < RC ? R - r0 : RC;
        const int s0 = r0 / JAC_SC * JAC_SC, ns = R - s0 < JAC_SC ? R - s0 : JAC_SC;      // this chunk's super-chunk: rows s0 .. s0 + ns
        if (r0 == s0) {
            AID_TRY(copy_f32(z_all + (size_t)s0 * L, zall, (size_t)ns * L, s));
            for (int j = 0; j < 4; ++j) {
                hipLaunchKernelGGL(perturb_rows_kernel, dim3((ns + 63) / 64), dim3(64), 0, s, z_all + (size_t)s0 * L, delta + ((size_t)j * R + s0) * L, e.perturbation_scale,
                                   zall + (size_t)(1 + j) * ns * L, ns, L);
                AID_LAUNCH_CHECK();
            }
            AID_TRY(conv_decoder_latent(dec, 5 * ns, zall, big, ws, s));
        }
        AID_TRY(conv_decoder_trunk(dec, prep, n, big + (size_t)(r0 - s0) * bigw, fz, ws, s));
        for (int j = 0; j < 4; ++j) {
            AID_TRY(conv_decoder_trunk(dec, prep, n, big + ((size_t)(1 + j) * ns + (r0 - s0)) * bigw, fp, ws, s));
            L1D(fd_to_cnhw_kernel, (size_t)3 * n * 168 * 168, s, fp, fz, e.perturbation_scale, dimg, 3, n, 168 * 168);
            // pixel_processor: 168 -> 84 -> 42 -> 21
            const PlainConv* pcs[3] = {&p0, &p2, &p4}; const float* bs[3] = {e.pp0_b, e.pp2_b, e.pp4_b};
            const float* xin = dimg; float* outs[3] = {c1, c2, c3}; int hi = 168;
            for (int l = 0; l < 3; ++l) {
                const int ho = hi / 2; const size_t N = (size_t)n * ho * ho;
                L1D(im2col5s2_kernel, (size_t)pcs[l]->Kp * N, s, xin, col, pcs[l]->cin, n, hi, ho, pcs[l]->K, (size_t)pcs[l]->Kp * N);
                GemmArgs ga = gemm_args(pcs[l]->w_eff, pcs[l]->Kp, col, (int)N, outs[l], (int)N, pcs[l]->cout, (int)N, pcs[l]->Kp);
                ga.b_kmajor = 1;
                AID_TRY(gemm_f32(ga, s));
                L1D(bias_rows_act_kernel, (size_t)pcs[l]->cout * N, s, outs[l], bs[l], pcs[l]->cout, N, ACT_RELU);
                xin = outs[l]; hi = ho;
            }
            // aggregator: tokens + positional encoding, K/V projections, 16-query attention, out_proj, Linear-LN-GELU-Linear
            const size_t NT = (size_t)n * T;
            L1D(add_pos_kernel, (size_t)C * NT, s, c3, e.pos, C, n, T);
            {
                GemmArgs ga = gemm_args(e.in_proj_w + (size_t)C * C, C, c3, (int)NT, kv, (int)NT, 2 * C, (int)NT, C);    // rows C..3C of in_proj: Wk, Wv
                ga.b_kmajor = 1;
                AID_TRY(gemm_f32(ga, s));
                L1D(bias_rows_act_kernel, (size_t)2 * C * NT, s, kv, e.in_proj_b + C, 2 * C, NT, ACT_NONE);
            }
            const float* mk = nullptr;
            if (train) {
                if (attn_mask) mk = attn_mask + ((size_t)j * R + r0) * HEADS * NQ * T;
                else { AID_TRY(philox_fill(mask_buf, (size_t)n * HEADS * NQ * T, 2, 0.1f, seed, ctr, s)); ctr += ((size_t)n * HEADS * NQ * T + 3) / 4; mk = mask_buf; }
            }
            hipLaunchKernelGGL(agg_attention_kernel, dim3(n * HEADS), dim3(256), sizeof(float) * ((size_t)NQ * T + NQ * 16 + 64), s, kv, q, mk, att, C, n, T, HEADS);
            AID_LAUNCH_CHECK();
            AID_TRY(linear_fwd(att, C, e.att_out.w, e.att_out.b, ao, C, n * NQ, C, C, ACT_NONE, s));
            AID_TRY(linear_fwd(ao, NQ * C, e.agg0.w, e.agg0.b, a0, e.agg0.out, n, e.agg0.out, NQ * C, ACT_NONE, s));
            AID_TRY(ln_act(a0, e.agg0.out, e.agg_ln.g, e.agg_ln.b, ACT_GELU, nullptr, a1, e.agg0.out, nullptr, n, e.agg0.out, s));
            AID_TRY(linear_fwd(a1, e.agg0.out, e.agg3.w, e.agg3.b, jin + (size_t)r0 * 4 * e.agg3.out + (size_t)j * e.agg3.out, 4 * e.agg3.out, n, e.agg3.out,
                               e.agg0.out, ACT_NONE, s));
        }
    }
    ws.release(mark);
    return AID_OK;
}
size_t pixel_jacobian_ws_bytes(const AidConvDecoder& dec) {
    return sizeof(float) * (size_t)5 * JAC_SC * ((size_t)dec.hidden * 441 + 8 * (size_t)dec.hidden + dec.L + 64) + aid_conv_decoder_workspace(&dec, 16) + sizeof(float) * (size_t)16 * (3 * 168 * 168 * 3 + 32 * 84 * 84 + 64 * 42 * 42 + 128 * 441 * 3 + 96 * 84 * 84 + 800 * 42 * 42
                                                                               + 16 * 128 * 2 + 8 * 16 * 441 + 4096) + (4u << 20);
}
}  // namespace aid

extern "C" {
}  // extern "C"
