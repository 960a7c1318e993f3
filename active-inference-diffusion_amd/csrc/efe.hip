// Expected-Free-Energy horizon rollout + MINE epistemic estimator, forward and policy backward.
// Reference: aid/core/active_inference.py:314-396 (rollout), :398-418 + :940-1063 (estimator), :815-836 (EMA loss).
//
// MI355X-first restructuring (results unchanged): the K=10 trajectories start from the same latent and are
// independent, so they are batched into M = K*B rows per horizon step (2 560-row GEMMs instead of 256-row
// ones); the estimator's K*S*B = 25 600 sample rows per step run as large MFMA GEMMs; per-(k,t) MINE
// statistics become segmented reductions, and the running-mean EMA - the only order-dependent piece - is
// replayed afterwards in the reference's (k outer, t inner) order.  The epistemic value carries no gradient
// (computed under no_grad in the reference), so the backward is BPTT over H steps through
// policy -> dynamics -> {reward, value} with parameter gradients for the policy only.
#include "mlp.h"
#include "score_net.h"   // philox_fill

namespace aid {

__device__ __forceinline__ float wg_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < 64) { t = (threadIdx.x < (blockDim.x >> 6)) ? sh[threadIdx.x] : 0.f; t = wave_sum(t); }
    return t;
}

__global__ void reparam_kernel(const float* __restrict__ mean, const float* __restrict__ eps, float sd, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mean[i] + eps[i] * sd;
}
// z_all[(k*S+s)*B + b] = next_mean[k*B + b] + eps * std          (active_inference.py:1009-1014)
// lv: per-element log-variance of the mean rows (compute_epistemic_value's next_latent_logvar), or null => the constant `sd`
__global__ void mine_sample_kernel(const float* __restrict__ nm, const float* __restrict__ eps, float sd, const float* __restrict__ lv,
                                   float* __restrict__ z, int K, int S, int B, int L) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)K * S * B * L;
    if (i >= n) return;
    const int l = i % L; const size_t r = i / L;
    const int b = r % B; const int k = (r / B) / S;
    const size_t src = ((size_t)k * B + b) * L + l;
    z[i] = nm[src] + eps[i] * (lv ? expf(0.5f * lv[src]) : sd);
}
// The four perturbed copies of every sample row, interleaved: zp4[4 r + j] = z[r] + normalize(delta[j][r]) * eps_scale.  Row order
// (r major, j minor) is what makes the concatenated Jacobian features [R, 4 FE] fall out of ONE 4R-row pass of the feature extractor.
__global__ void perturb4_kernel(const float* __restrict__ z, const float* __restrict__ d /*[4][R][L]*/, const float* __restrict__ scale,
                                float* __restrict__ zp4, size_t R, int L) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * R) return;
    const size_t r = t >> 2; const int j = (int)(t & 3);
    const float* dj = d + ((size_t)j * R + r) * L;
    float n2 = 0.f;
    for (int l = 0; l < L; ++l) { const float v = dj[l]; n2 += v * v; }
    const float inv = scale[0] / fmaxf(sqrtf(n2), 1e-12f);
    for (int l = 0; l < L; ++l) zp4[t * L + l] = z[r * L + l] + dj[l] * inv;
}
// finite differences of the decoder outputs, in place on the perturbed rows: fp4[4 r + j] = (fp4[4 r + j] - fz[r]) / eps_scale   (:963-970)
__global__ void fd4_kernel(float* __restrict__ fp4, const float* __restrict__ fz, const float* __restrict__ scale, size_t R, int O) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * R * O) return;
    const size_t row = i / O; const int o = (int)(i % O);
    fp4[i] = (fp4[i] - fz[(row >> 2) * O + o]) / scale[0];
}
// zp = z + normalize(delta) * eps_scale (F.normalize: / max(||d||, 1e-12))   (:962)
__global__ void perturb_kernel(const float* __restrict__ z, const float* __restrict__ d, const float* __restrict__ scale,
                               float* __restrict__ zp, size_t R, int L) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float n2 = 0.f;
    for (int l = 0; l < L; ++l) { const float v = d[r * L + l]; n2 += v * v; }
    const float inv = scale[0] / fmaxf(sqrtf(n2), 1e-12f);
    for (int l = 0; l < L; ++l) zp[r * L + l] = z[r * L + l] + d[r * L + l] * inv;
}
__global__ void fd_kernel(const float* __restrict__ fp, const float* __restrict__ fz, const float* __restrict__ scale,
                          float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (fp[i] - fz[i]) / scale[0];
}
// marginal features: rows of each (k,s) block permuted (:1029-1039); copies `nc` columns
__global__ void perm_rows_kernel(const float* __restrict__ src, int lds, const int32_t* __restrict__ perms, float* __restrict__ dst,
                                 int ldd, size_t R, int B, int nc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * nc) return;
    const size_t r = i / nc; const int c = i % nc;
    const size_t blk = r / B;
    const size_t sr = blk * B + (size_t)perms[r];
    dst[r * ldd + c] = src[sr * lds + c];
}
// one workgroup per trajectory k: joint mean, log-mean-exp of the marginal statistics, t_exp, MI  (:1046-1053,828-836)
__global__ void mine_reduce_kernel(const float* __restrict__ tj, const float* __restrict__ tm, int n, float* __restrict__ out /*[K][4]*/) {
    __shared__ float sh[16];
    __shared__ float s_mx;
    const float* a = tj + (size_t)blockIdx.x * n; const float* m = tm + (size_t)blockIdx.x * n;
    float acc = 0.f, mx = -3.4e38f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { acc += a[i]; mx = fmaxf(mx, m[i]); }
    const float js = wg_sum(acc, sh);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { float v = sh[0]; for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = fmaxf(v, sh[w]); s_mx = v; }
    __syncthreads();
    const float gmx = s_mx;
    float e0 = 0.f, e1 = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { e0 += expf(m[i]); e1 += expf(m[i] - gmx); }
    const float es = wg_sum(e0, sh); const float es1 = wg_sum(e1, sh);
    if (threadIdx.x == 0) {
        const float joint = js / (float)n;
        const float tlog = logf(es / (float)n);                               // EMALoss.forward: x.exp().mean().log()
        const float texp = expf((gmx + logf(es1)) - logf((float)n));          // exp(logsumexp(x) - log n)
        out[4 * blockIdx.x + 0] = joint; out[4 * blockIdx.x + 1] = tlog; out[4 * blockIdx.x + 2] = texp;
        out[4 * blockIdx.x + 3] = joint - tlog;
    }
}
// traj_efe += gamma^t (ew*epi + pw*prag + cw*cons), prag = pw*r/tau + V, cons = -entropy, epi = clamp(mi_k, 0)
__global__ void efe_accum_kernel(const float* __restrict__ rout, const float* __restrict__ V, const float* __restrict__ ent,
                                 const float* __restrict__ mine /*[K][4] or null*/, float gt, float ew, float pw, float cw, float tau,
                                 float* __restrict__ acc, float* __restrict__ prag_out, float* __restrict__ cons_out, int M, int B, int first) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float prag = pw * (rout[2 * r] / tau) + V[r];
    const float cons = -ent[r];
    const float epi = mine ? fmaxf(mine[4 * (r / B) + 3], 0.f) : 0.f;
    const float step = ew * epi + pw * prag + cw * cons;
    acc[r] = (first ? 0.f : acc[r]) + gt * step;
    prag_out[r] = prag; cons_out[r] = cons;
}
__global__ void efe_final_kernel(const float* __restrict__ acc, const float* __restrict__ prag, const float* __restrict__ cons,
                                 const float* __restrict__ mine, int K, int B, float* __restrict__ efe, float* __restrict__ info) {
    __shared__ float sh[16];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < K; ++k) t += acc[(size_t)k * B + b] / (float)K;
        efe[b] = t; a0 += t;
    }
    for (int r = threadIdx.x; r < K * B; r += blockDim.x) { a1 += prag[r]; a2 += cons[r]; }
    const float s0 = wg_sum(a0, sh); const float s1 = wg_sum(a1, sh); const float s2 = wg_sum(a2, sh);
    if (threadIdx.x == 0) {
        info[0] = s0 / (float)B; info[2] = s1 / (float)(K * B); info[3] = s2 / (float)(K * B);
        float e = 0.f;
        if (mine) { for (int k = 0; k < K; ++k) e += fmaxf(mine[4 * k + 3], 0.f); e /= (float)K;
                    info[4] = mine[4 * (K - 1) + 3]; info[5] = mine[4 * (K - 1) + 0]; info[6] = mine[4 * (K - 1) + 1]; }
        info[1] = e;
    }
}
// EMA of ema_loss (:828-836) replayed in the reference order: k outer, t inner.  texp: [H][K][4] (slot 2)
__global__ void running_mean_kernel(const float* __restrict__ mine_all, int H, int K, float alpha, float* __restrict__ rm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float r = rm[0];
    for (int k = 0; k < K; ++k)
        for (int t = 0; t < H; ++t) {
            const float te = mine_all[((size_t)t * K + k) * 4 + 2];
            if (r == 0.f) r = te;
            else r = (alpha * te) + (float)((1.0 - (double)alpha) * (double)r);
        }
    rm[0] = r;
}
__global__ void fill_pair_kernel(float* __restrict__ p, float v0, float v1, int M) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < M) { p[2 * r] = v0; p[2 * r + 1] = v1; }
}
__global__ void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (a ? a[i] : 0.f) + b[i] + c[i];
}
__global__ void tile_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, size_t n) {   // dst[k] = src for k<K
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * K) dst[i] = src[i % n];
}
// random permutation of 0..B-1 per block: bitonic sort of (random key, index) in LDS.  B <= 2048.
__global__ void randperm_blocks_kernel(const float* __restrict__ keys, int32_t* __restrict__ perms, int B, int P2) {
    extern __shared__ float sm[];
    float* kk = sm; int32_t* ix = (int32_t*)(sm + P2);
    const size_t base = (size_t)blockIdx.x * B;
    for (int i = threadIdx.x; i < P2; i += blockDim.x) { kk[i] = i < B ? keys[base + i] : 3.0f; ix[i] = i; }
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool sw = up ? (kk[i] > kk[l] || (kk[i] == kk[l] && ix[i] > ix[l])) : (kk[i] < kk[l] || (kk[i] == kk[l] && ix[i] < ix[l]));
                    if (sw) { const float tk = kk[i]; kk[i] = kk[l]; kk[l] = tk; const int32_t ti = ix[i]; ix[i] = ix[l]; ix[l] = ti; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < B; i += blockDim.x) perms[base + i] = ix[i];
}

#define LAUNCH1D(kern, n, s, ...)                                                                         \
    do {                                                                                                  \
        if ((n) > 0) { hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s, __VA_ARGS__); AID_LAUNCH_CHECK(); } \
    } while (0)

// ---- MINE forward for one horizon step, all K trajectories: fills mine_out [K][4]
struct PixelBackend { const AidConvDecoder* dec; const AidPixelEpistemicNet* epi; const float* attn_mask; uint64_t seed, offset; };
// compute_jacobian_features (active_inference.py:940-993) for Rm rows: finite differences of the decoder along 4 given directions -> feature extractor (state) or
// 5x5 convs + attention aggregator (pixels) -> jacobian_projector.  z_all holds the rows in [0, Rm) and (state branch) room for their 4 Rm perturbed copies behind
// them; fz / f0 / f1 / jin / pj are the caller's scratch; out [Rm][ldout] gets jacobian_projector's jp4.out columns.
static int jacobian_block(const AidDecoderNet& dec, const AidEpistemicNet& e, int Rm, float* z_all, float* fz, float* f0, float* f1, float* jin, const float* delta,
                          const float* proj_mask, bool train, LlaSave& pj, float* out, int ldout, Workspace& ws, hipStream_t s, const PixelBackend* px) {
    const size_t R = (size_t)Rm; const int L = e.L, O = px ? 1 : e.O, FE = e.fe4.out;
    if (px) {   // pixel branch: ConvDecoder finite differences -> 5x5 convs -> attention aggregator (conv.hip)
        AID_TRY(pixel_jacobian_features(*px->dec, *px->epi, Rm, z_all, delta, px->attn_mask, train, px->seed, px->offset, jin, ws, s));
    } else {
        AID_REQUIRE(5 * R < (size_t)1 << 30, AID_E_SHAPE, "mine_forward: %zu sample rows", R);
        float* zp4 = z_all + R * L; float* fp4 = fz + R * O;
        LAUNCH1D(perturb4_kernel, 4 * R, s, z_all, delta, e.perturbation_scale, zp4, R, L);
        AID_TRY(decoder_fwd(dec, (int)(5 * R), z_all, L, nullptr, fz, O, ws, s));       // decoder in eval mode (:949)
        LAUNCH1D(fd4_kernel, 4 * R * O, s, fp4, fz, e.perturbation_scale, R, O);
        const int R4 = (int)(4 * R);
        AID_TRY(linear_fwd(fp4, O, e.fe0.w, e.fe0.b, f0, e.fe0.out, R4, e.fe0.out, O, ACT_RELU, s));
        AID_TRY(linear_fwd(f0, e.fe0.out, e.fe2.w, e.fe2.b, f1, e.fe2.out, R4, e.fe2.out, e.fe0.out, ACT_RELU, s));
        AID_TRY(linear_fwd(f1, e.fe2.out, e.fe4.w, e.fe4.b, jin, FE, R4, FE, e.fe2.out, ACT_NONE, s));   // row 4 r + j of [4R, FE] == columns j FE.. of row r of [R, 4 FE]
    }
    AID_TRY(lla_fwd(e.jp0, e.jp_ln, ACT_RELU, train ? proj_mask : nullptr, jin, 4 * FE, Rm, pj, s, nullptr, false));   // gradient-free: nothing saved
    AID_TRY(linear_fwd(pj.y, e.jp0.out, e.jp4.w, e.jp4.b, out, ldout, Rm, e.jp4.out, e.jp0.out, ACT_NONE, s));
    return AID_OK;
}

static int mine_forward(const AidDecoderNet& dec, const AidEpistemicNet& e, int K, int S, int B, const float* next_mean,
                        const float* z_eps, const float* delta, const float* proj_mask, const float* mine_mask, const int32_t* perms,
                        bool train, float* mine_out, Workspace& ws, hipStream_t s, const PixelBackend* px = nullptr,
                        const float* next_logvar = nullptr) {
    const int L = e.L, O = px ? 1 : e.O;
    const size_t R = (size_t)K * S * B;
    const int Rm = (int)R;
    const size_t mark = ws.mark();
    // state branch: the sample rows and their 4 perturbed copies go through the decoder as ONE 5R-row batch (rows [0, R): z, rows R + 4 r + j:
    // copy j of row r), and the four feature-extractor passes as ONE 4R-row batch: same per-row arithmetic, 128 000-row GEMMs instead of
    // five 25 600-row ones (no tail effect, a fifth of the launches)
    const size_t RB = px ? R : 5 * R;
    float* z_all = ws.f32(RB * L); float* fz = px ? nullptr : ws.f32(RB * O);
    float* f0 = px ? nullptr : ws.f32(4 * R * e.fe0.out); float* f1 = px ? nullptr : ws.f32(4 * R * e.fe2.out);
    float* jin = ws.f32(R * 4 * e.fe4.out);
    float* comb = ws.f32(R * (e.jp4.out + e.lp2.out)); float* comb_m = ws.f32(R * (e.jp4.out + e.lp2.out));
    float* l0 = ws.f32(R * e.lp0.out);
    float* m0 = ws.f32(R * e.mn0.out); float* m1 = ws.f32(R * e.mn3.out); float* tj = ws.f32(R); float* tm = ws.f32(R);
    LlaSave pj; lla_alloc(Rm, e.jp0.out, ws, pj);
    AID_WS_OK(ws);
    const float sd = sqrtf(0.1f);   // exp(0.5 * ln 0.1)  (predict_next_latent :463)
    LAUNCH1D(mine_sample_kernel, R * L, s, next_mean, z_eps, expf(0.5f * logf(0.1f)), next_logvar, z_all, K, S, B, L);
    (void)sd;
    const int FE = e.fe4.out, CW = e.jp4.out + e.lp2.out;
    AID_TRY(jacobian_block(dec, e, Rm, z_all, fz, f0, f1, jin, delta, proj_mask, train, pj, comb, CW, ws, s, px));
    AID_TRY(linear_fwd(z_all, L, e.lp0.w, e.lp0.b, l0, e.lp0.out, Rm, e.lp0.out, L, ACT_RELU, s));
    AID_TRY(linear_fwd(l0, e.lp0.out, e.lp2.w, e.lp2.b, comb + e.jp4.out, CW, Rm, e.lp2.out, e.lp0.out, ACT_NONE, s));
    const size_t MSZ = R * e.mn0.out;
    auto mine_net = [&](const float* x, const float* k0, const float* k1, float* out) -> int {
        AID_TRY(linear_fwd(x, CW, e.mn0.w, e.mn0.b, m0, e.mn0.out, Rm, e.mn0.out, CW, ACT_RELU, s, nullptr, nullptr, 0, train ? k0 : nullptr));
        AID_TRY(linear_fwd(m0, e.mn0.out, e.mn3.w, e.mn3.b, m1, e.mn3.out, Rm, e.mn3.out, e.mn0.out, ACT_RELU, s, nullptr, nullptr, 0, train ? k1 : nullptr));
        AID_TRY(linear_fwd(m1, e.mn3.out, e.mn6.w, e.mn6.b, out, 1, Rm, 1, e.mn3.out, ACT_NONE, s));
        return AID_OK;
    };
    AID_TRY(mine_net(comb, mine_mask, mine_mask + MSZ, tj));
    LAUNCH1D(perm_rows_kernel, R * e.jp4.out, s, comb, CW, perms, comb_m, CW, R, B, e.jp4.out);
    AID_TRY(copy_cols(comb + e.jp4.out, CW, comb_m + e.jp4.out, CW, Rm, e.lp2.out, s));
    AID_TRY(mine_net(comb_m, mine_mask + 2 * MSZ, mine_mask + 3 * MSZ, tm));
    hipLaunchKernelGGL(mine_reduce_kernel, dim3(K), dim3(1024), 0, s, tj, tm, S * B, mine_out);
    AID_LAUNCH_CHECK();
    ws.release(mark);
    return AID_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// MINE estimator training step (DiffusionActiveInference.train_epistemic_estimator, active_inference.py:420-445):
// loss = -(mean t_joint - EMALoss(t_marginal)); gradients for every estimator-owned parameter (feature extractor, projector,
// latent processor, MINE network, perturbation_scale); the decoder runs under no_grad in the reference and gets none.
// ---------------------------------------------------------------------------------------------------------------------------
// d t_joint = -1/R ; d t_marg = exp(t_marg) / (running_mean_new + 1e-6) / R   (EMALoss.backward :822-826, loss = -mi)
// the estimator returns clamp(mi, min=0) (:1063): the whole gradient is gated by mi >= 0
__global__ void mine_loss_grad_kernel(const float* __restrict__ tm, const float* __restrict__ rm, const float* __restrict__ red, float* __restrict__ dtj,
                                      float* __restrict__ dtm, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float gate = red[3] >= 0.f ? 1.f : 0.f;
    dtj[i] = -gate / (float)R;
    dtm[i] = gate * expf(tm[i]) / (rm[0] + 1e-6f) / (float)R;
}
// running mean EMA for ONE estimator call (ema_loss :828-836), then out = {mi, joint, tlog, loss}
__global__ void mine_train_finish_kernel(const float* __restrict__ red /*[4]: joint, tlog, texp, mi*/, float alpha, float* __restrict__ rm, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float te = red[2];
    rm[0] = (rm[0] == 0.f) ? te : alpha * te + (1.0f - alpha) * rm[0];
    out[0] = red[3]; out[1] = red[0]; out[2] = red[1]; out[3] = -fmaxf(red[3], 0.f);
}
// dcomb[blk*B + perm[r], c] += dm[r, c] for the first nc columns (the marginal pass saw row-permuted Jacobian features)
__global__ void perm_rows_scatter_add_kernel(const float* __restrict__ dm, int ldm, const int32_t* __restrict__ perms, float* __restrict__ dst, int ldd,
                                             size_t R, int B, int nc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * nc) return;
    const size_t r = i / nc; const int c = i % nc;
    const size_t blk = r / B;
    dst[(blk * B + (size_t)perms[r]) * ldd + c] += dm[r * ldm + c];     // perms is a bijection inside each block: no two r collide
}
__global__ void scale_out_kernel(float* __restrict__ dst, const float* __restrict__ acc, const float* __restrict__ eps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] += acc[0] / eps[0];
}

static size_t mine_ws_bytes(const AidDecoderNet& dec, int K, int S, int B) {
    const size_t R = (size_t)K * S * B;
    return decoder_ws_bytes(dec, (int)(5 * R)) + sizeof(float) * R * (size_t)(5 * (dec.L > 128 ? dec.L : 128) + 5 * dec.O + 4 * 128 + 4 * 256 + 512 + 2 * 384 + 128 + 1024 + 2 + 1024 + 16) + (1 << 16);
}

}  // namespace aid

using namespace aid;

extern "C" {

size_t aid_efe_step_workspace(const AidPolicyNet* pol, const AidDecoderNet* dec, int B, const AidEfeCfg* cfg) {
    if (!pol || !cfg || B <= 0 || (!dec && !cfg->skip_epistemic)) return 0;
    const size_t M = (size_t)cfg->K * B, R = M * cfg->S, D = pol->D, H = cfg->H;
    size_t n = dec ? mine_ws_bytes(*dec, (int)(H * cfg->K), cfg->S, B) : 0;     // the estimator runs over all H horizon steps in one pass
    n += sizeof(float) * H * M * (52 * D + 128 + 2 * pol->L);                   // per-step saved activations (policy, dynamics, reward, value) + the policy adjoints the deferred weight gradients read
    n += sizeof(float) * M * (24 * D + 64);                           // backward scratch
    n += sizeof(float) * 8 * (4 * D * D + 2 * (size_t)pol->mean0.out * D + 2 * (size_t)pol->A * pol->mean0.out + D * (size_t)pol->L + 16 * D) + 4096;      // partial tiles of the split policy-gradient contractions
    n += sizeof(float) * (H * (M * (pol->A + pol->L) + R * pol->L * 5 + R * 512 * 5) + R * H);   // Philox noise when not given
    n += sizeof(int32_t) * H * R + (1 << 20);
    return n;
}

int aid_efe_step(const AidPolicyNet* pol, const AidPolicyNet* pgrads, const AidDynamicsNet* dyn, const AidRewardNet* rew,
                 const AidValueNet* val, const AidDecoderNet* dec, const AidEpistemicNet* epi, int B, const float* latents,
                 const AidEfeNoise* noise, const AidEfeCfg* cfg, float* running_mean, float* efe_out, float* info, void* wsp,
                 size_t ws_bytes, void* stream) {
    return aid_efe_step_ex(pol, pgrads, dyn, rew, val, dec, epi, B, latents, noise, cfg, running_mean, efe_out, info, nullptr, wsp, ws_bytes, stream);
}

int aid_efe_step_ex(const AidPolicyNet* pol, const AidPolicyNet* pgrads, const AidDynamicsNet* dyn, const AidRewardNet* rew,
                    const AidValueNet* val, const AidDecoderNet* dec, const AidEpistemicNet* epi, int B, const float* latents,
                    const AidEfeNoise* noise, const AidEfeCfg* cfg, float* running_mean, float* efe_out, float* info, float* next_means_out,
                    void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(pol && dyn && rew && val && latents && noise && cfg && running_mean && efe_out && info, AID_E_NULL,
                "aid_efe_step: null argument");
    AID_REQUIRE(cfg->skip_epistemic || (dec && epi), AID_E_NULL, "aid_efe_step: decoder / estimator tables are required unless skip_epistemic");
    const int H = cfg->H, K = cfg->K, S = cfg->S, L = pol->L, A = pol->A;
    AID_REQUIRE(B > 0 && H > 0 && K > 0 && S > 0 && B <= 2048, AID_E_SHAPE, "aid_efe_step: B=%d H=%d K=%d S=%d", B, H, K, S);
    hipStream_t s = (hipStream_t)stream;
    const int M = K * B; const size_t R = (size_t)M * S;
    const bool train = cfg->train != 0, do_epi = !cfg->skip_epistemic;
    Workspace ws(wsp, ws_bytes);
    uint64_t ctr = cfg->offset;
    auto draw = [&](const float* given, size_t cnt, int kind, float p, const float** out) -> int {
        if (given) { *out = given; return AID_OK; }
        float* buf = ws.f32(cnt); AID_WS_OK(ws);
        AID_TRY(philox_fill(buf, cnt, kind, p, cfg->seed, ctr, s)); ctr += (cnt + 3) / 4; *out = buf; return AID_OK;
    };
    const float *pol_eps, *rep_eps, *z_eps = nullptr, *delta = nullptr, *pmask = nullptr, *mmask = nullptr;
    const int32_t* perms = noise->perms;
    AID_TRY(draw(noise->policy_eps, (size_t)H * M * A, 0, 0.f, &pol_eps));
    AID_TRY(draw(noise->reparam_eps, (size_t)H * M * L, 0, 0.f, &rep_eps));
    if (do_epi) {
        AID_TRY(draw(noise->mine_z_eps, (size_t)H * R * L, 0, 0.f, &z_eps));
        AID_TRY(draw(noise->mine_delta, (size_t)H * 4 * R * L, 0, 0.f, &delta));
        if (train) {
            AID_TRY(draw(noise->proj_mask, (size_t)H * R * epi->jp0.out, 2, 0.1f, &pmask));
            AID_TRY(draw(noise->mine_mask, (size_t)H * 4 * R * epi->mn0.out, 2, 0.1f, &mmask));
        }
        if (!perms) {
            const float* keys; AID_TRY(draw(nullptr, (size_t)H * R, 1, 0.f, &keys));
            int32_t* pb = (int32_t*)ws.raw(sizeof(int32_t) * (size_t)H * R); AID_WS_OK(ws);
            int P2 = 1; while (P2 < B) P2 <<= 1;
            hipLaunchKernelGGL(randperm_blocks_kernel, dim3((unsigned)(H * K * S)), dim3(256), (size_t)P2 * 8, s, keys, pb, B, P2);
            AID_LAUNCH_CHECK(); perms = pb;
        }
    }
    // ---- per-step saved state
    struct Step { float* cur; float* next_mean; float* nxt; PolicySave ps; DynSave ds; RewardSave rs; ValueSave vs; float* V; float* tvec; };
    Step st[16];
    AID_REQUIRE(H <= 16, AID_E_SHAPE, "aid_efe_step: horizon %d > 16", H);
    float* acc = ws.f32(M); float* prag = ws.f32(M); float* cons = ws.f32(M);
    float* mine_all = ws.f32((size_t)H * K * 4);
    // [H+1][K*B][L]: slot 0 = the K copies of the belief latents, slot t + 1 = the state reached after horizon step t.  Slots 0..H-1 are the
    // policy's inputs of the H forward passes, side by side: its first layer's weight gradient is one contraction over H*K*B rows.
    float* cur_all = ws.f32((size_t)(H + 1) * M * L);
    float* lat_rep = cur_all;
    PolicySteps psteps;
    const bool defer_grads = pgrads != nullptr;
    if (defer_grads) AID_TRY(policy_steps_alloc(*pol, M, H, ws, psteps));
    float* next_mean_all = ws.f32((size_t)H * M * L);      // [H][K*B][L]: the estimator runs ONCE over the dynamics means of every horizon step
    AID_WS_OK(ws);
    LAUNCH1D(tile_rows_kernel, (size_t)M * L, s, latents, lat_rep, K, (size_t)B * L);   // latent.clone() per trajectory
    const float sd = expf(0.5f * logf(0.1f));
    const float* cur = lat_rep;
    for (int t = 0; t < H; ++t) {
        Step& q = st[t];
        q.cur = (float*)cur; q.next_mean = next_mean_all + (size_t)t * M * L; q.nxt = cur_all + (size_t)(t + 1) * M * L; q.V = ws.f32(M); q.tvec = ws.f32(M);
        if (defer_grads) q.ps = psteps.sv[t];
        AID_WS_OK(ws);
        AID_TRY(policy_fwd(*pol, M, cur, L, pol_eps + (size_t)t * M * A, q.ps, ws, s));
        AID_TRY(dynamics_fwd(*dyn, M, cur, L, q.ps.action, A, q.next_mean, q.ds, ws, s));
        if (next_means_out) AID_TRY(copy_f32(q.next_mean, next_means_out + (size_t)t * M * L, (size_t)M * L, s));
        LAUNCH1D(reparam_kernel, (size_t)M * L, s, q.next_mean, rep_eps + (size_t)t * M * L, sd, q.nxt, (size_t)M * L);
        AID_TRY(reward_fwd(*rew, M, q.nxt, L, q.rs, ws, s));
        AID_TRY(fill_const(q.tvec, (float)t, M, s));
        AID_TRY(value_fwd(*val, M, q.nxt, L, q.tvec, q.V, q.vs, ws, s));
        cur = q.nxt;
    }
    // The epistemic term is gradient-free and feeds nothing the rollout reads, so it is evaluated after the rollout, for all H steps at
    // once: H*K trajectory segments = 128 000 sample rows (640 000 through the decoder) per GEMM instead of 25 600 five times - no
    // partial last round of workgroups (25 600 rows are 1.56 rounds of the 64-row kernels), a fifth of the launches, and the rollout's
    // small dependent kernels run back to back (next to the ELBO chain of the other stream) instead of between the big GEMMs.  The MINE
    // statistics stay per (k, t) segment and the running mean is replayed in the reference's (k outer, t inner) order below.
    if (do_epi)
        AID_TRY(mine_forward(*dec, *epi, H * K, S, B, next_mean_all, z_eps, delta, train ? pmask : nullptr, train ? mmask : nullptr, perms, train, mine_all, ws, s));
    for (int t = 0; t < H; ++t) {
        Step& q = st[t];
        LAUNCH1D(efe_accum_kernel, M, s, q.rs.out, q.V, q.ps.ent, do_epi ? mine_all + (size_t)t * K * 4 : nullptr, powf(cfg->discount, (float)t), cfg->epistemic_weight,
                 cfg->pragmatic_weight, cfg->consistency_weight, cfg->temperature, acc, prag, cons, M, B, t == 0);
    }
    hipLaunchKernelGGL(efe_final_kernel, dim3(1), dim3(1024), 0, s, acc, prag, cons, do_epi ? mine_all + (size_t)(H - 1) * K * 4 : nullptr, K, B, efe_out, info);
    AID_LAUNCH_CHECK();
    if (do_epi) { hipLaunchKernelGGL(running_mean_kernel, dim3(1), dim3(64), 0, s, mine_all, H, K, cfg->mine_alpha, running_mean); AID_LAUNCH_CHECK(); }
    if (!pgrads) return AID_OK;

    // ---- backward (BPTT): policy_loss = mean_b efe_b
    float* d_r = ws.f32((size_t)M * 2); float* d_v = ws.f32(M); float* entc = ws.f32(M);
    float* dz_r = ws.f32((size_t)M * L); float* dz_v = ws.f32((size_t)M * L); float* dnxt = ws.f32((size_t)M * L);
    float* dz_dyn = ws.f32((size_t)M * L); float* dz_pol = ws.f32((size_t)M * L); float* da = ws.f32((size_t)M * A);
    float* carry = ws.f32((size_t)M * L);
    AID_WS_OK(ws);
    bool have_carry = false;
    for (int t = H - 1; t >= 0; --t) {
        Step& q = st[t];
        const float c = powf(cfg->discount, (float)t) * cfg->pragmatic_weight / ((float)K * (float)B);   // dLoss/dprag
        LAUNCH1D(fill_pair_kernel, M, s, d_r, c * cfg->pragmatic_weight / cfg->temperature, 0.f, M);
        AID_TRY(fill_const(d_v, c, M, s));
        AID_TRY(fill_const(entc, -powf(cfg->discount, (float)t) * cfg->consistency_weight / ((float)K * (float)B), M, s));
        AID_TRY(reward_bwd_data(*rew, M, q.nxt, L, q.rs, d_r, dz_r, L, ws, s));
        AID_TRY(value_bwd(*val, nullptr, M, q.tvec, q.vs, d_v, dz_v, L, ws, s));
        LAUNCH1D(add3_kernel, (size_t)M * L, s, have_carry ? carry : nullptr, dz_r, dz_v, dnxt, (size_t)M * L);
        AID_TRY(dynamics_bwd(*dyn, nullptr, M, q.ds, dnxt, dz_dyn, da, ws, s));
        AID_TRY(policy_bwd(*pol, pgrads, M, q.cur, L, pol_eps + (size_t)t * M * A, q.ps, da, entc, t > 0 ? dz_pol : nullptr, L, ws, s, &psteps, t));
        if (t > 0) { LAUNCH1D(add3_kernel, (size_t)M * L, s, (const float*)nullptr, dz_dyn, dz_pol, carry, (size_t)M * L); have_carry = true; }
    }
    // the policy's weight / bias gradients: dW = sum_t adj_t^T x_t as ONE contraction per layer over the H*K*B rows of the [H][K*B][.] buffers,
    // all nine layers in one grouped launch (45 launches inside the BPTT chain before)
    GradQueue gq(s);
    {   // room for the partial tiles of the long contractions (8 row chunks of every policy gradient)
        const size_t Dp = pol->D, Hh = pol->mean0.out;
        gq.scratch_floats = 8 * (4 * Dp * Dp + 2 * Hh * Dp + 2 * (size_t)pol->A * Hh + Dp * (size_t)L + 16 * Dp);
        gq.scratch = ws.f32(gq.scratch_floats); AID_WS_OK(ws);
    }
    AID_TRY(policy_bptt_grads(*pol, *pgrads, psteps, cur_all, L, gq, ws));
    AID_TRY(gq.flush());
    return AID_OK;
}


size_t aid_mine_train_workspace(const AidDecoderNet* dec, const AidDynamicsNet* dyn, int B, int S) {
    if (!dec || !dyn || B <= 0 || S <= 0) return 0;
    const size_t R = (size_t)S * B;
    // decoder scratch + saved activations of every trainable stage (4 feature-extractor passes, projector, latent processor, two MINE
    // passes) + their gradients + Philox noise when none is given + dynamics forward
    return decoder_ws_bytes(*dec, (int)R) + sizeof(float) * R * (size_t)(16384 + 12 * dec->O + 8 * dec->L) + sizeof(float) * (size_t)B * (40 * dyn->D + 64)
           + sizeof(int32_t) * R + (1 << 20);
}

int aid_mine_train_step(const AidDynamicsNet* dyn, const AidDecoderNet* dec, const AidEpistemicNet* epi, const AidEpistemicNet* grads, int B, int S,
                        const float* latents, const float* actions, const AidEfeNoise* noise, int train, uint64_t seed, uint64_t offset,
                        float alpha, float* running_mean, float* out4, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dyn && dec && epi && grads && latents && actions && noise && running_mean && out4, AID_E_NULL, "aid_mine_train_step: null argument");
    AID_REQUIRE(B > 0 && S > 0 && B <= 2048, AID_E_SHAPE, "aid_mine_train_step: B=%d S=%d", B, S);
    hipStream_t s = (hipStream_t)stream;
    const AidEpistemicNet& e = *epi; const AidEpistemicNet& ge = *grads;
    const int L = e.L, O = e.O, Rm = S * B, FE = e.fe4.out, JW = e.jp4.out, CW = e.jp4.out + e.lp2.out, MH = e.mn0.out;
    const size_t R = (size_t)Rm;
    Workspace ws(wsp, ws_bytes);
    uint64_t ctr = offset;
    auto draw = [&](const float* given, size_t cnt, int kind, float p, const float** out) -> int {
        if (given) { *out = given; return AID_OK; }
        float* buf = ws.f32(cnt); AID_WS_OK(ws);
        AID_TRY(philox_fill(buf, cnt, kind, p, seed, ctr, s)); ctr += (cnt + 3) / 4; *out = buf; return AID_OK;
    };
    const float *z_eps, *delta, *pmask = nullptr, *mmask = nullptr;
    AID_TRY(draw(noise->mine_z_eps, R * L, 0, 0.f, &z_eps));
    AID_TRY(draw(noise->mine_delta, 4 * R * L, 0, 0.f, &delta));
    if (train) {
        AID_TRY(draw(noise->proj_mask, R * e.jp0.out, 2, 0.1f, &pmask));
        AID_TRY(draw(noise->mine_mask, 4 * R * MH, 2, 0.1f, &mmask));
    }
    const int32_t* perms = noise->perms;
    if (!perms) {
        const float* keys; AID_TRY(draw(nullptr, R, 1, 0.f, &keys));
        int32_t* pb = (int32_t*)ws.raw(sizeof(int32_t) * R); AID_WS_OK(ws);
        int P2 = 1; while (P2 < B) P2 <<= 1;
        hipLaunchKernelGGL(randperm_blocks_kernel, dim3((unsigned)S), dim3(256), (size_t)P2 * 8, s, keys, pb, B, P2);
        AID_LAUNCH_CHECK(); perms = pb;
    }
    // ---- forward, everything trainable saved
    float* next_mean = ws.f32((size_t)B * L);
    DynSave dsv;
    AID_TRY(dynamics_fwd(*dyn, B, latents, L, actions, dyn->A, next_mean, dsv, ws, s));
    float* z_all = ws.f32(R * L); float* zp = ws.f32(R * L); float* fz = ws.f32(R * O); float* fp = ws.f32(R * O);
    float* diff[4]; LaSave f0[4], f1[4];
    for (int j = 0; j < 4; ++j) { diff[j] = ws.f32(R * O); la_alloc(Rm, e.fe0.out, true, ws, f0[j]); la_alloc(Rm, e.fe2.out, true, ws, f1[j]); }
    float* jin = ws.f32(R * 4 * FE);
    LlaSave pj; lla_alloc(Rm, e.jp0.out, ws, pj);
    float* comb = ws.f32(R * CW); float* comb_m = ws.f32(R * CW);
    LaSave l0; la_alloc(Rm, e.lp0.out, true, ws, l0);
    LaSave mj0, mj1, mm0, mm1;
    la_alloc(Rm, MH, true, ws, mj0); la_alloc(Rm, e.mn3.out, true, ws, mj1); la_alloc(Rm, MH, true, ws, mm0); la_alloc(Rm, e.mn3.out, true, ws, mm1);
    float* tj = ws.f32(R); float* tm = ws.f32(R); float* red = ws.f32(8);
    AID_WS_OK(ws);
    LAUNCH1D(mine_sample_kernel, R * L, s, next_mean, z_eps, expf(0.5f * logf(0.1f)), (const float*)nullptr, z_all, 1, S, B, L);
    AID_TRY(decoder_fwd(*dec, Rm, z_all, L, nullptr, fz, O, ws, s));                     // decoder: eval mode, no gradient (:949-953)
    for (int j = 0; j < 4; ++j) {
        LAUNCH1D(perturb_kernel, R, s, z_all, delta + (size_t)j * R * L, e.perturbation_scale, zp, R, L);
        AID_TRY(decoder_fwd(*dec, Rm, zp, L, nullptr, fp, O, ws, s));
        LAUNCH1D(fd_kernel, R * O, s, fp, fz, e.perturbation_scale, diff[j], R * O);
        AID_TRY(la_fwd(e.fe0, ACT_RELU, nullptr, diff[j], O, Rm, f0[j], e.fe0.out, s));
        AID_TRY(la_fwd(e.fe2, ACT_RELU, nullptr, f0[j].y, e.fe0.out, Rm, f1[j], e.fe2.out, s));
        AID_TRY(linear_fwd(f1[j].y, e.fe2.out, e.fe4.w, e.fe4.b, jin + (size_t)j * FE, 4 * FE, Rm, FE, e.fe2.out, ACT_NONE, s));
    }
    AID_TRY(lla_fwd(e.jp0, e.jp_ln, ACT_RELU, train ? pmask : nullptr, jin, 4 * FE, Rm, pj, s));
    AID_TRY(linear_fwd(pj.y, e.jp0.out, e.jp4.w, e.jp4.b, comb, CW, Rm, JW, e.jp0.out, ACT_NONE, s));
    AID_TRY(la_fwd(e.lp0, ACT_RELU, nullptr, z_all, L, Rm, l0, e.lp0.out, s));
    AID_TRY(linear_fwd(l0.y, e.lp0.out, e.lp2.w, e.lp2.b, comb + JW, CW, Rm, e.lp2.out, e.lp0.out, ACT_NONE, s));
    const size_t MSZ = R * MH;
    auto mine_fwd = [&](const float* x, const float* k0, const float* k1, LaSave& a0, LaSave& a1, float* out) -> int {
        AID_TRY(la_fwd(e.mn0, ACT_RELU, train ? k0 : nullptr, x, CW, Rm, a0, MH, s));
        AID_TRY(la_fwd(e.mn3, ACT_RELU, train ? k1 : nullptr, a0.y, MH, Rm, a1, e.mn3.out, s));
        AID_TRY(linear_fwd(a1.y, e.mn3.out, e.mn6.w, e.mn6.b, out, 1, Rm, 1, e.mn3.out, ACT_NONE, s));
        return AID_OK;
    };
    AID_TRY(mine_fwd(comb, mmask, mmask ? mmask + MSZ : nullptr, mj0, mj1, tj));
    LAUNCH1D(perm_rows_kernel, R * JW, s, comb, CW, perms, comb_m, CW, R, B, JW);
    AID_TRY(copy_cols(comb + JW, CW, comb_m + JW, CW, Rm, e.lp2.out, s));
    AID_TRY(mine_fwd(comb_m, mmask ? mmask + 2 * MSZ : nullptr, mmask ? mmask + 3 * MSZ : nullptr, mm0, mm1, tm));
    hipLaunchKernelGGL(mine_reduce_kernel, dim3(1), dim3(1024), 0, s, tj, tm, Rm, red);
    hipLaunchKernelGGL(mine_train_finish_kernel, dim3(1), dim3(64), 0, s, red, alpha, running_mean, out4);
    AID_LAUNCH_CHECK();
    // ---- backward
    float* dtj = ws.f32(R); float* dtm = ws.f32(R);
    float* d1 = ws.f32(R * e.mn3.out); float* d0 = ws.f32(R * MH);
    float* dcomb = ws.f32(R * CW); float* dcm = ws.f32(R * CW);
    AID_WS_OK(ws);
    hipLaunchKernelGGL(mine_loss_grad_kernel, dim3((Rm + 255) / 256), dim3(256), 0, s, tm, running_mean, red, dtj, dtm, Rm);
    AID_LAUNCH_CHECK();
    auto mine_bwd = [&](const float* x, const float* k0, const float* k1, const LaSave& a0, const LaSave& a1, float* dt, float* dx) -> int {
        AID_TRY(linear_bwd_weight(dt, 1, a1.y, e.mn3.out, ge.mn6.w, Rm, 1, e.mn3.out, s, 1));
        AID_TRY(colsum(dt, 1, ge.mn6.b, Rm, 1, 1, s, &ws));
        AID_TRY(linear_bwd_data(dt, 1, e.mn6.w, e.mn3.out, d1, e.mn3.out, Rm, 1, s, 0));
        AID_TRY(la_bwd(e.mn3, &ge.mn3, ACT_RELU, train ? k1 : nullptr, a0.y, MH, Rm, a1, d1, e.mn3.out, d0, MH, s, 0, &ws));
        AID_TRY(la_bwd(e.mn0, &ge.mn0, ACT_RELU, train ? k0 : nullptr, x, CW, Rm, a0, d0, MH, dx, CW, s, 0, &ws));
        return AID_OK;
    };
    AID_TRY(mine_bwd(comb, mmask, mmask ? mmask + MSZ : nullptr, mj0, mj1, dtj, dcomb));
    AID_TRY(mine_bwd(comb_m, mmask ? mmask + 2 * MSZ : nullptr, mmask ? mmask + 3 * MSZ : nullptr, mm0, mm1, dtm, dcm));
    LAUNCH1D(perm_rows_scatter_add_kernel, R * JW, s, dcm, CW, perms, dcomb, CW, R, B, JW);
    AID_TRY(axpby_cols(dcomb + JW, CW, 1.0f, dcm + JW, CW, 1.0f, dcomb + JW, CW, Rm, e.lp2.out, s));
    // latent processor
    {
        float* dl = ws.f32(R * e.lp2.out); float* dl0 = ws.f32(R * e.lp0.out); AID_WS_OK(ws);
        AID_TRY(copy_cols(dcomb + JW, CW, dl, e.lp2.out, Rm, e.lp2.out, s));
        AID_TRY(linear_bwd_weight(dl, e.lp2.out, l0.y, e.lp0.out, ge.lp2.w, Rm, e.lp2.out, e.lp0.out, s, 1));
        AID_TRY(colsum(dl, e.lp2.out, ge.lp2.b, Rm, e.lp2.out, 1, s, &ws));
        AID_TRY(linear_bwd_data(dl, e.lp2.out, e.lp2.w, e.lp0.out, dl0, e.lp0.out, Rm, e.lp2.out, s, 0));
        AID_TRY(la_bwd(e.lp0, &ge.lp0, ACT_RELU, nullptr, z_all, L, Rm, l0, dl0, e.lp0.out, nullptr, 0, s, 0, &ws));
    }
    // projector
    float* djin = ws.f32(R * 4 * FE);
    {
        float* dj = ws.f32(R * JW); float* dpy = ws.f32(R * e.jp0.out); AID_WS_OK(ws);
        AID_TRY(copy_cols(dcomb, CW, dj, JW, Rm, JW, s));
        AID_TRY(linear_bwd_weight(dj, JW, pj.y, e.jp0.out, ge.jp4.w, Rm, JW, e.jp0.out, s, 1));
        AID_TRY(colsum(dj, JW, ge.jp4.b, Rm, JW, 1, s, &ws));
        AID_TRY(linear_bwd_data(dj, JW, e.jp4.w, e.jp0.out, dpy, e.jp0.out, Rm, JW, s, 0));
        AID_TRY(lla_bwd(e.jp0, e.jp_ln, &ge.jp0, &ge.jp_ln, ACT_RELU, train ? pmask : nullptr, jin, 4 * FE, Rm, pj, dpy, djin, 4 * FE, ws, s));
    }
    // feature extractor x4 and the perturbation scale (diff = (f(z+d) - f(z)) / eps is the only place eps carries a gradient)
    {
        float* dfe = ws.f32(R * FE); float* df1 = ws.f32(R * e.fe2.out); float* df0 = ws.f32(R * e.fe0.out); float* ddiff = ws.f32(R * O);
        float* acc = ws.f32(8); AID_WS_OK(ws);
        AID_CHECK_HIP(hipMemsetAsync(acc, 0, 8 * sizeof(float), s));
        for (int j = 0; j < 4; ++j) {
            AID_TRY(copy_cols(djin + (size_t)j * FE, 4 * FE, dfe, FE, Rm, FE, s));
            AID_TRY(linear_bwd_weight(dfe, FE, f1[j].y, e.fe2.out, ge.fe4.w, Rm, FE, e.fe2.out, s, 1));
            AID_TRY(colsum(dfe, FE, ge.fe4.b, Rm, FE, 1, s, &ws));
            AID_TRY(linear_bwd_data(dfe, FE, e.fe4.w, e.fe2.out, df1, e.fe2.out, Rm, FE, s, 0));
            AID_TRY(la_bwd(e.fe2, &ge.fe2, ACT_RELU, nullptr, f0[j].y, e.fe0.out, Rm, f1[j], df1, e.fe2.out, df0, e.fe0.out, s, 0, &ws));
            AID_TRY(la_bwd(e.fe0, &ge.fe0, ACT_RELU, nullptr, diff[j], O, Rm, f0[j], df0, e.fe0.out, ddiff, O, s, 0, &ws));
            AID_TRY(dot_accum(ddiff, diff[j], R * O, -1.0f, acc, s));                    // d eps += -sum(ddiff * diff) / eps
        }
        hipLaunchKernelGGL(scale_out_kernel, dim3(1), dim3(64), 0, s, ge.perturbation_scale, acc, e.perturbation_scale);
        AID_LAUNCH_CHECK();
    }
    return AID_OK;
}

size_t aid_epistemic_value_workspace(const AidDecoderNet* dec, int B, int S) {
    if (!dec || B <= 0 || S <= 0) return 0;
    const size_t R = (size_t)S * B;
    return mine_ws_bytes(*dec, 1, S, B) + sizeof(float) * R * (size_t)(5 * dec->L + 5 * 512 + 1) + sizeof(int32_t) * R + (1 << 20);
}

int aid_epistemic_value(const AidDecoderNet* dec, const AidEpistemicNet* epi, int B, int S, const float* next_mean, const float* next_logvar,
                        const AidEfeNoise* noise, int train, uint64_t seed, uint64_t offset, float alpha, float* running_mean, float* out4,
                        void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dec && epi && next_mean && next_logvar && noise && running_mean && out4, AID_E_NULL, "aid_epistemic_value: null argument");
    AID_REQUIRE(B > 0 && S > 0 && B <= 2048, AID_E_SHAPE, "aid_epistemic_value: B=%d S=%d", B, S);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const AidEpistemicNet& e = *epi;
    const size_t R = (size_t)S * B; const int L = e.L;
    uint64_t ctr = offset;
    auto draw = [&](const float* given, size_t cnt, int kind, float p, const float** out) -> int {
        if (given) { *out = given; return AID_OK; }
        float* buf = ws.f32(cnt); AID_WS_OK(ws);
        AID_TRY(philox_fill(buf, cnt, kind, p, seed, ctr, s)); ctr += (cnt + 3) / 4; *out = buf; return AID_OK;
    };
    const float *z_eps, *delta, *pmask = nullptr, *mmask = nullptr;
    AID_TRY(draw(noise->mine_z_eps, R * L, 0, 0.f, &z_eps));
    AID_TRY(draw(noise->mine_delta, 4 * R * L, 0, 0.f, &delta));
    if (train) {
        AID_TRY(draw(noise->proj_mask, R * e.jp0.out, 2, 0.1f, &pmask));
        AID_TRY(draw(noise->mine_mask, 4 * R * e.mn0.out, 2, 0.1f, &mmask));
    }
    const int32_t* perms = noise->perms;
    if (!perms) {
        const float* keys; AID_TRY(draw(nullptr, R, 1, 0.f, &keys));
        int32_t* pb = (int32_t*)ws.raw(sizeof(int32_t) * R); AID_WS_OK(ws);
        int P2 = 1; while (P2 < B) P2 <<= 1;
        hipLaunchKernelGGL(randperm_blocks_kernel, dim3((unsigned)S), dim3(256), (size_t)P2 * 8, s, keys, pb, B, P2);
        AID_LAUNCH_CHECK(); perms = pb;
    }
    float* red = ws.f32(8); AID_WS_OK(ws);
    AID_TRY(mine_forward(*dec, e, 1, S, B, next_mean, z_eps, delta, pmask, mmask, perms, train != 0, red, ws, s, nullptr, next_logvar));
    // running-mean EMA of this one call, then out4 = {mi, joint, marginal term, -clamp(mi, 0)}: the caller negates slot 3
    hipLaunchKernelGGL(mine_train_finish_kernel, dim3(1), dim3(64), 0, s, red, alpha, running_mean, out4);
    AID_LAUNCH_CHECK();
    return AID_OK;
}

size_t aid_pixel_epistemic_workspace(const AidConvDecoder* dec, int K, int S, int B);
// FunctionSpaceEpistemicEstimator.compute_jacobian_features (active_inference.py:940-993) as a call of its own: z [R][L] -> out [R][jp4.out].
// delta [4][R][L]: the four standard-normal perturbation directions (normalised and scaled inside, F.normalize(randn) * epsilon :962; null => drawn from
// (seed, offset));
// proj_mask [R][jp0.out]: jacobian_projector's dropout mask in train mode (null => drawn).  State branch: `dec` is the observation decoder; pixel branch
// (aid_pixel_jacobian_features): the ConvDecoder + the estimator's pixel processor and spatial aggregator.
size_t aid_jacobian_features_workspace(const AidDecoderNet* dec, int R) {
    if (!dec || R <= 0) return 0;
    return mine_ws_bytes(*dec, 1, 1, R) + sizeof(float) * (size_t)R * (size_t)(5 * dec->L + 512) + (1 << 20);
}
static int jacobian_entry(const AidDecoderNet& dec, const AidEpistemicNet& e, int R, const float* z, const float* delta, const float* proj_mask, int train,
                          uint64_t seed, uint64_t offset, float* out, Workspace& ws, hipStream_t s, const PixelBackend* px) {
    const int L = e.L, O = px ? 1 : e.O;
    const size_t Rz = (size_t)R, RB = px ? Rz : 5 * Rz;
    uint64_t ctr = offset;
    if (!delta) {
        float* d = ws.f32(4 * Rz * L); AID_WS_OK(ws);
        AID_TRY(philox_fill(d, 4 * Rz * L, 0, 0.f, seed, ctr, s)); ctr += (4 * Rz * L + 3) / 4;
        delta = d;
    }
    if (train && !proj_mask) {
        float* m = ws.f32(Rz * e.jp0.out); AID_WS_OK(ws);
        AID_TRY(philox_fill(m, Rz * e.jp0.out, 2, 0.1f, seed, ctr, s)); ctr += (Rz * e.jp0.out + 3) / 4;
        proj_mask = m;
    }
    float* z_all = ws.f32(RB * L); float* fz = px ? nullptr : ws.f32(RB * O);
    float* f0 = px ? nullptr : ws.f32(4 * Rz * e.fe0.out); float* f1 = px ? nullptr : ws.f32(4 * Rz * e.fe2.out);
    float* jin = ws.f32(Rz * 4 * e.fe4.out);
    LlaSave pj; lla_alloc(R, e.jp0.out, ws, pj);
    AID_WS_OK(ws);
    AID_CHECK_HIP(hipMemcpyAsync(z_all, z, sizeof(float) * Rz * L, hipMemcpyDeviceToDevice, s));
    PixelBackend pxl; if (px) { pxl = *px; pxl.offset = ctr + (1ull << 36); }
    return jacobian_block(dec, e, R, z_all, fz, f0, f1, jin, delta, proj_mask, train != 0, pj, out, e.jp4.out, ws, s, px ? &pxl : nullptr);
}
int aid_jacobian_features(const AidDecoderNet* dec, const AidEpistemicNet* epi, int R, const float* z, const float* delta, const float* proj_mask, int train,
                          uint64_t seed, uint64_t offset, float* out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dec && epi && z && out, AID_E_NULL, "aid_jacobian_features: null argument");
    AID_REQUIRE(R > 0 && R <= (1 << 24), AID_E_SHAPE, "aid_jacobian_features: R=%d", R);
    Workspace ws(wsp, ws_bytes);
    return jacobian_entry(*dec, *epi, R, z, delta, proj_mask, train, seed, offset, out, ws, (hipStream_t)stream, nullptr);
}
size_t aid_pixel_jacobian_features_workspace(const AidConvDecoder* dec, int R) { return aid_pixel_epistemic_workspace(dec, 1, 1, R); }
int aid_pixel_jacobian_features(const AidConvDecoder* dec, const AidPixelEpistemicNet* pe, int R, const float* z, const float* delta, const float* proj_mask,
                                const float* attn_mask, int train, uint64_t seed, uint64_t offset, float* out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dec && pe && z && out, AID_E_NULL, "aid_pixel_jacobian_features: null argument");
    AID_REQUIRE(R > 0 && R <= (1 << 20), AID_E_SHAPE, "aid_pixel_jacobian_features: R=%d", R);
    Workspace ws(wsp, ws_bytes);
    AidEpistemicNet e = {};
    e.L = pe->L; e.O = 1; e.perturbation_scale = pe->perturbation_scale;
    e.fe0.out = 1; e.fe2.out = 1; e.fe4.out = pe->agg3.out;
    e.jp0 = pe->jp0; e.jp_ln = pe->jp_ln; e.jp4 = pe->jp4;
    PixelBackend px = {dec, pe, attn_mask, seed, offset};
    AidDecoderNet none = {};
    return jacobian_entry(none, e, R, z, delta, proj_mask, train, seed, offset, out, ws, (hipStream_t)stream, &px);
}

size_t aid_pixel_epistemic_workspace(const AidConvDecoder* dec, int K, int S, int B) {
    if (!dec || K <= 0 || S <= 0 || B <= 0) return 0;
    const size_t R = (size_t)K * S * B;
    return pixel_jacobian_ws_bytes(*dec) + sizeof(float) * R * (size_t)(2 * dec->L + 4 * 256 + 2 * 384 + 128 + 3 * 512 + 8) + (4u << 20);
}

int aid_pixel_epistemic_value(const AidConvDecoder* dec, const AidPixelEpistemicNet* pe, int K, int S, int B, const float* next_mean,
                              const AidEfeNoise* noise, const float* attn_mask, int train, uint64_t seed, uint64_t offset, float* mine_out,
                              void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dec && pe && next_mean && noise && mine_out, AID_E_NULL, "aid_pixel_epistemic_value: null argument");
    AID_REQUIRE(K > 0 && S > 0 && B > 0 && B <= 2048, AID_E_SHAPE, "aid_pixel_epistemic_value: K=%d S=%d B=%d", K, S, B);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const size_t R = (size_t)K * S * B; const int L = pe->L;
    // the tail (projector, latent processor, MINE statistics) is the state branch's: view the pixel table through its struct
    AidEpistemicNet e = {};
    e.L = L; e.O = 1; e.perturbation_scale = pe->perturbation_scale;
    e.fe0.out = 1; e.fe2.out = 1; e.fe4.out = pe->agg3.out;
    e.jp0 = pe->jp0; e.jp_ln = pe->jp_ln; e.jp4 = pe->jp4; e.lp0 = pe->lp0; e.lp2 = pe->lp2; e.mn0 = pe->mn0; e.mn3 = pe->mn3; e.mn6 = pe->mn6;
    uint64_t ctr = offset;
    auto draw = [&](const float* given, size_t cnt, int kind, float p, const float** out) -> int {
        if (given) { *out = given; return AID_OK; }
        float* buf = ws.f32(cnt); AID_WS_OK(ws);
        AID_TRY(philox_fill(buf, cnt, kind, p, seed, ctr, s)); ctr += (cnt + 3) / 4; *out = buf; return AID_OK;
    };
    const float *z_eps, *delta, *pmask = nullptr, *mmask = nullptr;
    AID_TRY(draw(noise->mine_z_eps, R * L, 0, 0.f, &z_eps));
    AID_TRY(draw(noise->mine_delta, 4 * R * L, 0, 0.f, &delta));
    if (train) {
        AID_TRY(draw(noise->proj_mask, R * e.jp0.out, 2, 0.1f, &pmask));
        AID_TRY(draw(noise->mine_mask, 4 * R * e.mn0.out, 2, 0.1f, &mmask));
    }
    const int32_t* perms = noise->perms;
    if (!perms) {
        const float* keys; AID_TRY(draw(nullptr, R, 1, 0.f, &keys));
        int32_t* pb = (int32_t*)ws.raw(sizeof(int32_t) * R); AID_WS_OK(ws);
        int P2 = 1; while (P2 < B) P2 <<= 1;
        hipLaunchKernelGGL(randperm_blocks_kernel, dim3((unsigned)(K * S)), dim3(256), (size_t)P2 * 8, s, keys, pb, B, P2);
        AID_LAUNCH_CHECK(); perms = pb;
    }
    PixelBackend px = {dec, pe, attn_mask, seed, ctr + (1ull << 36)};
    AidDecoderNet none = {};
    return mine_forward(none, e, K, S, B, next_mean, z_eps, delta, pmask, mmask, perms, train != 0, mine_out, ws, s, &px);
}

__global__ void efe_add_epistemic_kernel(const float* __restrict__ mine_all, int H, int K, int B, float ew, float disc, float* __restrict__ efe,
                                         float* __restrict__ info) {
    float add = 0.f, g = 1.f;
    for (int t = 0; t < H; ++t) {
        float e = 0.f;
        for (int k = 0; k < K; ++k) e += fmaxf(mine_all[((size_t)t * K + k) * 4 + 3], 0.f);
        add += g * ew * e / (float)K; g *= disc;
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) efe[b] += add;
    if (threadIdx.x == 0) {
        info[0] += add;                                       // policy_loss = mean_b efe_b
        float e = 0.f;                                        // epistemic_mean: mean over trajectories of the LAST horizon step's value (:381-389)
        for (int k = 0; k < K; ++k) e += fmaxf(mine_all[((size_t)(H - 1) * K + k) * 4 + 3], 0.f);
        info[1] = e / (float)K;
        const float* last = mine_all + ((size_t)(H - 1) * K + (K - 1)) * 4;
        info[4] = last[3]; info[5] = last[0]; info[6] = last[1];
    }
}

int aid_efe_add_epistemic(const float* mine_all, int H, int K, int B, float epistemic_weight, float discount, float alpha, float* running_mean,
                          float* efe, float* info, void* stream) {
    AID_REQUIRE(mine_all && running_mean && efe && info && H > 0 && K > 0 && B > 0, AID_E_NULL, "aid_efe_add_epistemic: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(efe_add_epistemic_kernel, dim3(1), dim3(256), 0, s, mine_all, H, K, B, epistemic_weight, discount, efe, info);
    hipLaunchKernelGGL(running_mean_kernel, dim3(1), dim3(64), 0, s, mine_all, H, K, alpha, running_mean);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
}  // extern "C"
