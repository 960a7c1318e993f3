// compute_diffusion_elbo forward+backward (aid/core/active_inference.py:533-636, 709-771), belief update
// (:256-312), decode_observation (:217-242) and the fused clip -> AdamW -> EMA optimiser step
// (aid/agents/state_agent.py:152-158).  Host orchestration + the pointwise / reduction kernels.
#include "score_net.h"
#include "score_bwd.h"
#include "rowwise_bwd.h"
#include "mlp.h"

namespace aid {

__device__ __forceinline__ float blk_sum(float v, float* sh) {   // sum over a <=1024-thread workgroup; valid in thread 0
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < 64) { t = (threadIdx.x < (blockDim.x >> 6)) ? sh[threadIdx.x] : 0.f; t = wave_sum(t); }
    return t;
}

// z_t = sqrt(alpha) z + sqrt(sigma) eps,  alpha = sigmoid(lambda), sigma = sigmoid(-lambda)   (diffusion.py:56-91)
__global__ void elbo_prepare_kernel(const float* __restrict__ z, const float* __restrict__ t, const float* __restrict__ eps,
                                    const float* __restrict__ lmin, const float* __restrict__ lmax, float* __restrict__ zt, int B, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L;
    const float lam = lmin[0] + (lmax[0] - lmin[0]) * (1.f - t[b]);
    const float alpha = 1.f / (1.f + expf(-lam)), sigma = 1.f / (1.f + expf(lam));
    zt[i] = sqrtf(alpha) * z[i] + sqrtf(sigma) * eps[i];
}

// per-sample score-matching loss, its gradient wrt the predicted score, and the direct part of d(loss)/d(lambda_b)
__global__ void elbo_score_loss_kernel(const float* __restrict__ score, const float* __restrict__ eps, const float* __restrict__ t,
                                       const float* __restrict__ lmin, const float* __restrict__ lmax, float dw, float* __restrict__ per,
                                       float* __restrict__ wout, float* __restrict__ d_score, float* __restrict__ dlamA, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float tb = t[b];
    const float lam = lmin[0] + (lmax[0] - lmin[0]) * (1.f - tb);
    const float alpha = 1.f / (1.f + expf(-lam)), sigma = 1.f / (1.f + expf(lam));
    const float w = expf(-0.5f * (lam * lam) / 4.0f) * (sinf(tb * 3.14159265358979323846f) + 0.1f);   // diffusion.py:93-104
    const float inv = 1.f / (sigma + 1e-8f);
    float ss = 0.f, cross = 0.f;
    for (int l = 0; l < L; ++l) {
        const float e = eps[(size_t)b * L + l];
        const float diff = score[(size_t)b * L + l] + e * inv;          // s - s*,  s* = -eps/(sigma+1e-8)
        ss += diff * diff;
        cross += 2.f * diff * (e * sigma * alpha * inv * inv);           // 2 diff * (-ds*/dlambda)
        d_score[(size_t)b * L + l] = -dw * 2.f * w * diff / (float)B;    // loss = -elbo
    }
    per[b] = w * ss; wout[b] = w;
    dlamA[b] = (-dw / (float)B) * ((-lam / 4.0f) * w * ss + w * cross);
}

__global__ void elbo_lambda_grad_kernel(const float* __restrict__ dlamA, const float* __restrict__ dz, const float* __restrict__ z,
                                        const float* __restrict__ eps, const float* __restrict__ t, const float* __restrict__ lmin,
                                        const float* __restrict__ lmax, float* __restrict__ dlmin, float* __restrict__ dlmax, int B, int L) {
    __shared__ float sh[16];
    float a0 = 0.f, a1 = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float tb = t[b];
        const float lam = lmin[0] + (lmax[0] - lmin[0]) * (1.f - tb);
        const float alpha = 1.f / (1.f + expf(-lam)), sigma = 1.f / (1.f + expf(lam));
        const float ca = 0.5f * sqrtf(alpha) * sigma, cs = -0.5f * sqrtf(sigma) * alpha;   // d sqrt(alpha)/dlam, d sqrt(sigma)/dlam
        float d = dlamA[b];
        for (int l = 0; l < L; ++l) d += dz[(size_t)b * L + l] * (ca * z[(size_t)b * L + l] + cs * eps[(size_t)b * L + l]);
        a0 += d * tb; a1 += d * (1.f - tb);
    }
    const float s0 = blk_sum(a0, sh); const float s1 = blk_sum(a1, sh);
    if (threadIdx.x == 0) { dlmin[0] += s0; dlmax[0] += s1; }
}

// KL-ish prior term (diffusion.py:146-152, active_inference.py:466-476,609-611) + gradients of the learnable prior
__global__ void kl_prior_kernel(const float* __restrict__ z, const float* __restrict__ peps, const float* __restrict__ pm,
                                const float* __restrict__ pls, const float* __restrict__ t, float kcfg, float* __restrict__ dpm,
                                float* __restrict__ dpls, float* __restrict__ out /*kl, klw, mean_t*/, int B, int L,
                                const float* __restrict__ mean_t_global) {
    __shared__ float sh[16];
    __shared__ float s_klw;
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) acc += t[b];
    const float st = blk_sum(acc, sh);
    // data parallel: the annealing weight exp(-5 mean t) (:611) is a statistic of the GLOBAL batch, handed in by the caller
    if (threadIdx.x == 0) { const float mt = mean_t_global ? mean_t_global[0] : st / (float)B; s_klw = expf(-5.0f * mt); out[1] = s_klw; out[2] = mt; }
    __syncthreads();
    const float c = kcfg * s_klw / (float)B;
    acc = 0.f;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const int l = i % L;
        const float d = z[i] - (pm[l] + expf(pls[l]) * peps[i]);
        acc += 0.5f * d * d;
    }
    const float kl = blk_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = kl / (float)B;
    if (dpm) {      // one wave per latent dim, lanes over the batch, fixed-order butterfly: deterministic
        const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
        for (int l = threadIdx.x >> 6; l < L; l += nw) {
            float g0 = 0.f, g1 = 0.f;
            const float sd = expf(pls[l]);
            for (int b = lane; b < B; b += 64) {
                const float e = peps[(size_t)b * L + l];
                const float d = z[(size_t)b * L + l] - (pm[l] + sd * e);
                g0 += d; g1 += d * sd * e;
            }
            g0 = wave_sum(g0); g1 = wave_sum(g1);
            if (lane == 0) { dpm[l] += c * g0; dpls[l] += c * g1; }
        }
    }
}

__global__ void mse_scalar_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int M, int N, float* __restrict__ out) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < M * N; i += blockDim.x) { const float d = a[(size_t)(i / N) * lda + i % N] - b[(size_t)(i / N) * ldb + i % N]; acc += d * d; }
    const float s = blk_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = s / (float)(M * N);
}
int mse_scalar(const float* a, int lda, const float* b, int ldb, int M, int N, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mse_scalar_kernel, dim3(1), dim3(1024), 0, s, a, lda, b, ldb, M, N, out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void reward_nll_kernel(const float* __restrict__ rp, const float* __restrict__ r, int B, float* __restrict__ out) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float m = rp[2 * b], ls = fminf(fmaxf(rp[2 * b + 1], -5.f), 2.f), sd = expf(ls);
        const float d = r[b] - m;
        acc += (d * d) / (2.f * sd * sd) + logf(sd) + 0.9189385332046727f;
    }
    const float s = blk_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = s / (float)B;
}

__global__ void mean_scalar_kernel(const float* __restrict__ a, int n, float* __restrict__ out) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += a[i];
    const float s = blk_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

// metrics: 0 recon 1 kl 2 sm 3 elbo 4 nll 5 gp 6 mean_t 7 w_mean 8 loss ; tmp: [0]=kl [1]=klw [2]=mean_t
__global__ void elbo_metrics_kernel(float* __restrict__ m, const float* __restrict__ tmp, float kcfg, float dw, float rw, float gpw) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    m[1] = tmp[0]; m[6] = tmp[2];
    const float elbo = -m[0] + kcfg * tmp[0] * tmp[1] + dw * m[2] + gpw * m[5] - rw * m[4];
    m[3] = elbo; m[8] = -elbo;
}

// sequential EMA of active_inference.py:750-771 (python-double arithmetic, float32 storage), order-dependent by design
// (an update touches one bin and bins do not interact: thread b applies, in sample order, the updates of bin b - the same arithmetic in the same
// order per bin as the sequential loop, without its 256 dependent round trips to memory: 44 -> ~5 us.  Launch with >= 100 threads.)
__global__ void time_importance_update_kernel(float* __restrict__ w, const float* __restrict__ t, const float* __restrict__ per, int B, int first) {
    __shared__ int sb[1024];
    __shared__ float sp[1024];
    if (blockIdx.x != 0) return;
    const int tid = threadIdx.x;
    float wv = tid < 100 ? (first ? 1.0f : w[tid]) : 0.f;
    for (int c0 = 0; c0 < B; c0 += 1024) {
        const int nc = B - c0 < 1024 ? B - c0 : 1024;
        __syncthreads();
        for (int i = tid; i < nc; i += blockDim.x) {
            int bin = (int)(long long)(t[c0 + i] * 99.0f);
            sb[i] = bin < 0 ? 0 : (bin > 99 ? 99 : bin); sp[i] = per[c0 + i];
        }
        __syncthreads();
        if (tid < 100)
            for (int i = 0; i < nc; ++i)
                if (sb[i] == tid) wv = (float)(0.99 * (double)wv + 0.01 * (double)sp[i]);
    }
    if (tid < 100) w[tid] = wv;
}

// importance sampling of t (active_inference.py:731-748): softmax(w) -> inverse-CDF categorical -> (idx + u)/100
__global__ void sample_time_kernel(const float* __restrict__ w, int first, const float* __restrict__ u1, const float* __restrict__ u2,
                                   float* __restrict__ t, int B) {
    __shared__ float cdf[100];
    if (threadIdx.x == 0 && !first) {
        float mx = w[0];
        for (int i = 1; i < 100; ++i) mx = fmaxf(mx, w[i]);
        float s = 0.f;
        for (int i = 0; i < 100; ++i) { s += expf(w[i] - mx); cdf[i] = s; }
        for (int i = 0; i < 100; ++i) cdf[i] /= s;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (first) { t[b] = u2[b]; continue; }
        int idx = 0;
        while (idx < 99 && cdf[idx] <= u1[b]) ++idx;
        t[b] = ((float)idx + u2[b]) / 100.0f;
    }
}

// latent batch statistics of update_belief_via_diffusion (:286-291): mean and unbiased std over the batch
__global__ void batch_stats_kernel(const float* __restrict__ z, int B, int L, float* __restrict__ out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += z[(size_t)b * L + l];
    const float m = s / (float)B;
    float v = 0.f;
    for (int b = 0; b < B; ++b) { const float d = z[(size_t)b * L + l] - m; v += d * d; }
    out[l] = B == 1 ? z[l] : m;
    out[L + l] = B > 1 ? sqrtf(v / (float)(B - 1)) : 0.f;
}

// ---------------------------------------------------------------------------------------------- optimiser
constexpr int OPT_BLOCKS = 1024;
__global__ void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float sh[16];
    float acc = 0.f;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = ((((uintptr_t)g) & 15) == 0) ? n / 4 : 0;       // 16-B loads when the arena is aligned; fixed order either way
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (size_t i = tid; i < n4; i += nthr) { const f4 v = reinterpret_cast<const f4*>(g)[i]; acc += v[0] * v[0]; acc += v[1] * v[1]; acc += v[2] * v[2]; acc += v[3] * v[3]; }
    for (size_t i = n4 * 4 + tid; i < n; i += nthr) { const float v = g[i]; acc += v * v; }
    const float s = blk_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void finalize_norm_kernel(const float* __restrict__ partial, int nb, float clip, float* __restrict__ out, const float* __restrict__ ext) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
    const float s = blk_sum(acc, sh);
    if (threadIdx.x == 0) {
        out[2] = s;                                       // this arena's own sum of squares
        const float nrm = sqrtf(s + (ext ? ext[0] : 0.f));   // + the other arenas of the same clip group (pixel mode: score + encoder)
        out[0] = nrm;
        out[1] = clip > 0.f ? fminf(clip / (nrm + 1e-6f), 1.0f) : 1.0f;   // torch clip_grad_norm_
    }
}
// torch.optim.AdamW single-tensor update + EMAModel.update, one pass: 5 reads + 4 writes per parameter (36 B).  16-B accesses; the moments,
// the EMA copy and the gradient are touched once per step, so they go through non-temporal loads / stores and leave the Infinity Cache
// to the parameters the next forward pass re-reads.
__device__ __forceinline__ void adamw_one(float& pi, float gi, float& mi, float& vi, float* emai, float c, float lr, float wd, float b1, float b2,
                                          float eps, float step_size, float sqrt_bc2, float ema_decay, int coupled_l2) {
    gi *= c;
    if (coupled_l2) gi += wd * pi;            // torch.optim.Adam(weight_decay=wd): L2 term folded into the gradient
    else pi *= (1.f - lr * wd);                // torch.optim.AdamW: decoupled decay
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi = pi - step_size * (mi / denom);
    if (emai) *emai = ema_decay * *emai + (1.f - ema_decay) * pi;
}
typedef float opt_f4 __attribute__((ext_vector_type(4)));
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             float* __restrict__ ema, size_t n, float lr, float wd, float b1, float b2, float eps, float bc1,
                             float sqrt_bc2, const float* __restrict__ coef, float ema_decay, int coupled_l2, int vec) {
    const float c = coef[1];
    const float step_size = lr / bc1;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = tid; i < n4; i += nthr) {
        opt_f4 pv = reinterpret_cast<opt_f4*>(p)[i];
        const opt_f4 gv = __builtin_nontemporal_load(reinterpret_cast<const opt_f4*>(g) + i);
        opt_f4 mv = __builtin_nontemporal_load(reinterpret_cast<const opt_f4*>(m) + i);
        opt_f4 vv = __builtin_nontemporal_load(reinterpret_cast<const opt_f4*>(v) + i);
        opt_f4 ev = {0.f, 0.f, 0.f, 0.f};
        if (ema) ev = __builtin_nontemporal_load(reinterpret_cast<const opt_f4*>(ema) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pv[j], mj = mv[j], vj = vv[j], ej = ev[j];
            adamw_one(pj, gv[j], mj, vj, ema ? &ej : nullptr, c, lr, wd, b1, b2, eps, step_size, sqrt_bc2, ema_decay, coupled_l2);
            pv[j] = pj; mv[j] = mj; vv[j] = vj; ev[j] = ej;
        }
        reinterpret_cast<opt_f4*>(p)[i] = pv;
        __builtin_nontemporal_store(mv, reinterpret_cast<opt_f4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<opt_f4*>(v) + i);
        if (ema) __builtin_nontemporal_store(ev, reinterpret_cast<opt_f4*>(ema) + i);
    }
    for (size_t i = n4 * 4 + tid; i < n; i += nthr) {
        float pi = p[i], mi = m[i], vi = v[i], ei = ema ? ema[i] : 0.f;
        adamw_one(pi, g[i], mi, vi, ema ? &ei : nullptr, c, lr, wd, b1, b2, eps, step_size, sqrt_bc2, ema_decay, coupled_l2);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema) ema[i] = ei;
    }
}

}  // namespace aid

using namespace aid;

extern "C" {

int aid_grad_sumsq(const float* g, size_t n, float* scratch, void* stream) {
    AID_REQUIRE(g && scratch, AID_E_NULL, "aid_grad_sumsq: null");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)((n + 255) / 256 > OPT_BLOCKS ? OPT_BLOCKS : (n + 255) / 256);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, scratch + 8);
    hipLaunchKernelGGL(finalize_norm_kernel, dim3(1), dim3(1024), 0, s, scratch + 8, nb, 0.f, scratch, (const float*)nullptr);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_adamw_step_ext(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay, float beta1,
                       float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch, const float* ext_sumsq, void* stream);

int aid_adamw_step(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay, float beta1,
                   float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch, void* stream) {
    return aid_adamw_step_ext(p, g, m, v, ema, n, lr, weight_decay, beta1, beta2, eps, step, clip_norm, ema_decay, scratch, nullptr, stream);
}

static int adam_step_impl(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay, float beta1,
                          float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch, const float* ext_sumsq, int coupled_l2,
                          void* stream);

int aid_adam_l2_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float weight_decay, float beta1, float beta2, float eps,
                     int step, float clip_norm, float* scratch, void* stream) {
    return adam_step_impl(p, g, m, v, nullptr, n, lr, weight_decay, beta1, beta2, eps, step, clip_norm, 0.f, scratch, nullptr, 1, stream);
}

int aid_adamw_step_ext(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay, float beta1,
                       float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch, const float* ext_sumsq, void* stream) {
    return adam_step_impl(p, g, m, v, ema, n, lr, weight_decay, beta1, beta2, eps, step, clip_norm, ema_decay, scratch, ext_sumsq, 0, stream);
}

static int adam_step_impl(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay, float beta1,
                          float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch, const float* ext_sumsq, int coupled_l2,
                          void* stream) {
    AID_REQUIRE(p && g && m && v && scratch, AID_E_NULL, "aid_adamw_step: null arena");
    AID_REQUIRE(step >= 1, AID_E_SHAPE, "aid_adamw_step: step must be 1-based");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return AID_OK;
    ProfScope prof(PROF_OPTIM, s, 0.0, (double)n * 4.0 * (ema ? 10.0 : 8.0));
    const int nb = (int)((n + 255) / 256 > OPT_BLOCKS ? OPT_BLOCKS : (n + 255) / 256);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, scratch + 8);
    hipLaunchKernelGGL(finalize_norm_kernel, dim3(1), dim3(1024), 0, s, scratch + 8, nb, clip_norm, scratch, ext_sumsq);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) == 0) ? 1 : 0;
    const size_t work = vec ? (n + 3) / 4 : n;
    const int ub = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
    hipLaunchKernelGGL(adamw_kernel, dim3(ub), dim3(256), 0, s, p, g, m, v, ema, n, lr, weight_decay, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), scratch, ema_decay, coupled_l2, vec);
    AID_LAUNCH_CHECK();
    return AID_OK;
}

int aid_time_importance_update(float* time_importance, const float* t, const float* per_sample, int n, int first_call, void* stream) {
    AID_REQUIRE(time_importance && t && per_sample, AID_E_NULL, "aid_time_importance_update: null tensor");
    hipLaunchKernelGGL(time_importance_update_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, time_importance, t, per_sample, n, first_call);
    AID_LAUNCH_CHECK(); return AID_OK;
}

size_t aid_decode_workspace(const AidDecoderNet* dec, int B) { return dec && B > 0 ? decoder_ws_bytes(*dec, B) + 1024 : 0; }

int aid_decode_observation(const AidDecoderNet* dec, int B, const float* z, const float* mask0, const float* mask1, const float* mask2,
                           float* out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dec && z && out, AID_E_NULL, "aid_decode_observation: null argument");
    Workspace ws(wsp, ws_bytes);
    const float* masks[3] = {mask0, mask1, mask2};
    return decoder_fwd(*dec, B, z, dec->L, (mask0 || mask1 || mask2) ? masks : nullptr, out, dec->O, ws, (hipStream_t)stream);
}

size_t aid_belief_update_workspace(const AidScoreNet* net, const AidDecoderNet* dec, int B, int n_steps) {
    if (!net || !dec) return 0;
    const size_t a = aid_denoise_loop_workspace(net, B, n_steps);
    const size_t b = decoder_ws_bytes(*dec, B) + sizeof(float) * (size_t)B * (dec->O + 2 * dec->s[0].out + dec->s[2].out) + 4096;
    return (a > b ? a : b) + 4096;
}

int aid_belief_update(const AidScoreNet* net, const AidDecoderNet* dec, int B, int n_steps, const float* coefs, const float* obs,
                      const float* z_init, const float* step_noise, const float* drop_masks, const float* const* dec_masks, int train,
                      uint64_t seed, uint64_t offset, float* z_out, float* stats_out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(net && dec && z_out && stats_out, AID_E_NULL, "aid_belief_update: null argument");
    hipStream_t s = (hipStream_t)stream;
    AID_TRY(aid_denoise_loop(net, B, n_steps, coefs, obs, z_init, step_noise, drop_masks, train, seed, offset, z_out, nullptr, wsp, ws_bytes, stream));
    const int L = net->L, O = dec->O;
    Workspace ws(wsp, ws_bytes);
    float* rec = ws.f32((size_t)B * O);
    const float* masks[3] = {nullptr, nullptr, nullptr};
    float* mbuf[3] = {nullptr, nullptr, nullptr};
    if (train) {
        uint64_t ctr = offset + ((uint64_t)1 << 40);   // decoder masks live in their own counter range
        for (int i = 0; i < 3; ++i) {
            if (dec_masks && dec_masks[i]) masks[i] = dec_masks[i];
            else {
                const size_t n = (size_t)B * dec->s[i].out;
                mbuf[i] = ws.f32(n); AID_WS_OK(ws);
                AID_TRY(philox_fill(mbuf[i], n, 2, 0.2f, seed, ctr, s)); ctr += (n + 3) / 4; masks[i] = mbuf[i];
            }
        }
    }
    AID_WS_OK(ws);
    AID_TRY(decoder_fwd(*dec, B, z_out, L, train ? masks : nullptr, rec, O, ws, s));
    hipLaunchKernelGGL(batch_stats_kernel, dim3((L + 63) / 64), dim3(64), 0, s, z_out, B, L, stats_out);
    AID_LAUNCH_CHECK();
    AID_TRY(mse_scalar(rec, O, obs, O, B, O, stats_out + 2 * L, s));
    return AID_OK;
}

size_t aid_elbo_step_workspace(const AidScoreNet* net, const AidDecoderNet* dec, int B) {
    if (!net || B <= 0) return 0;
    const size_t D = net->D, L = net->L;
    size_t n = 2 * score_acts_bytes(*net, B, true) + score_tan_bytes(*net, B) + score_reverse_ws_bytes(*net, B) +
               score_cond_backward_ws_bytes(*net, B) + (dec ? decoder_ws_bytes(*dec, B) : 0);
    n += sizeof(float) * (size_t)B * ((size_t)(2 * net->n_blocks + 1) * 2 * D + 12 * L + 2 * D + (dec ? dec->O + 2 * dec->s[0].out + dec->s[2].out : net->O) + 8 * D + 64);
    return n + (1 << 16) + SplitScope::bytes();
}

int aid_mse(const float* a, int lda, const float* b, int ldb, int M, int N, float* out, void* stream) {
    AID_REQUIRE(a && b && out, AID_E_NULL, "aid_mse: null argument");
    AID_REQUIRE(M > 0 && N > 0 && lda >= N && ldb >= N, AID_E_SHAPE, "aid_mse: M=%d N=%d lda=%d ldb=%d", M, N, lda, ldb);
    return mse_scalar(a, lda, b, ldb, M, N, out, (hipStream_t)stream);
}

int aid_gradient_penalty(const AidScoreNet* net, int B, const float* z, const float* t, int t_mode, const float* obs, const float* drop_mask,
                         float* penalty_out, float* grad_out, void* wsp, size_t ws_bytes, void* stream) {
    AID_TRY(score_validate(net));
    AID_REQUIRE(z && t && obs && penalty_out, AID_E_NULL, "aid_gradient_penalty: null argument");
    AID_REQUIRE(B > 0, AID_E_SHAPE, "aid_gradient_penalty: B=%d", B);
    AID_REQUIRE(t_mode == AID_T_CONTINUOUS || t_mode == AID_T_DISCRETE, AID_E_SHAPE, "aid_gradient_penalty: t_mode=%d", t_mode);
    hipStream_t s = (hipStream_t)stream;
    const AidScoreNet& n = *net;
    const int D = n.D, L = n.L;
    Workspace ws(wsp, ws_bytes);
    float* gvec = grad_out ? grad_out : ws.f32((size_t)B * L);
    float* wvec = ws.f32((size_t)B * L);
    ScoreActs ag;
    AID_TRY(score_acts_alloc(n, B, true, ws, ag)); AID_WS_OK(ws);
    AID_TRY(score_time_embed(n, B, t, t_mode, ag, s));
    AID_TRY(score_obs_stage0(n, B, obs, ag, s));
    AID_TRY(score_cond(n, B, drop_mask, ag.temb, D, ag, s));
    AID_TRY(score_trunk(n, B, z, ag, true, s));
    AID_TRY(score_reverse(n, nullptr, B, ag, nullptr, z, nullptr, nullptr, t, t_mode, 0.f, gvec, nullptr, ws, s));   // g = d(sum s)/dz
    return gp_weights(gvec, wvec, penalty_out, B, L, s);
}

// LatentScoreNetwork as a differentiable op: forward (activations kept) + reverse pass for a caller-supplied cotangent d_score = dLoss/dscore.
// Parameter gradients are ACCUMULATED into `grads` (every weight / bias gradient of the pass in one grouped launch, as in aid_elbo_step, whose main pass this is
// without the loss in between); dz_out (optional) receives dLoss/dz, d_obs_out (optional) ACCUMULATES dLoss/dobs.  What `loss.backward()` does for a loss built
// on `score_network(z, t, obs)` in the reference (models/score_networks.py:101-171 under autograd).  Workspace: aid_elbo_step_workspace(net, NULL, B).
int aid_score_backward(const AidScoreNet* net, const AidScoreNet* grads, int B, const float* z, const float* t, int t_mode, const float* obs, const float* drop_mask,
                       const float* d_score, float* score_out, float* dz_out, float* d_obs_out, void* wsp, size_t ws_bytes, void* stream) {
    AID_TRY(score_validate(net)); AID_TRY(score_validate(grads));
    AID_REQUIRE(z && t && obs && d_score, AID_E_NULL, "aid_score_backward: null argument");
    AID_REQUIRE(B > 0, AID_E_SHAPE, "aid_score_backward: B=%d", B);
    AID_REQUIRE(t_mode == AID_T_CONTINUOUS || t_mode == AID_T_DISCRETE, AID_E_SHAPE, "aid_score_backward: t_mode=%d", t_mode);
    hipStream_t s = (hipStream_t)stream;
    const AidScoreNet& n = *net;
    const int D = n.D, L = n.L, NA = 2 * n.n_blocks + 1, ldmod = NA * 2 * D;
    Workspace ws(wsp, ws_bytes);
    SplitScope split(ws, s);
    float* score = score_out ? score_out : ws.f32((size_t)B * L);
    float* dz = dz_out ? dz_out : ws.f32((size_t)B * L);
    float* dmod = ws.f32((size_t)B * ldmod);
    AID_WS_OK(ws);
    ScoreActs am;
    AID_TRY(score_acts_alloc(n, B, true, ws, am));
    AID_TRY(score_time_embed(n, B, t, t_mode, am, s));
    AID_TRY(score_obs_stage0(n, B, obs, am, s));
    AID_TRY(score_cond(n, B, drop_mask, am.temb, D, am, s));
    AID_TRY(score_trunk(n, B, z, am, true, s));
    AID_TRY(score_head(am.raw, n.output_multiplier, t, t_mode, score, B, L, s));
    GradQueue gq(s);
    AID_TRY(score_reverse(n, grads, B, am, nullptr, z, nullptr, d_score, t, t_mode, 0.f, dz, dmod, ws, s, &gq));
    AID_TRY(score_cond_backward(n, *grads, B, am, dmod, obs, drop_mask, t, t_mode, ws, s, d_obs_out, &gq));
    AID_TRY(gq.flush());
    return AID_OK;
}

int aid_sample_time(const float* time_importance, int first_call, int B, uint64_t seed, uint64_t offset, float* t_out, void* wsp,
                    size_t ws_bytes, void* stream) {
    AID_REQUIRE(time_importance && t_out, AID_E_NULL, "aid_sample_time: null argument");
    AID_REQUIRE(B > 0, AID_E_SHAPE, "aid_sample_time: B=%d", B);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    float* u1 = ws.f32(B); float* u2 = ws.f32(B); AID_WS_OK(ws);
    AID_TRY(philox_fill(u1, B, 1, 0.f, seed, offset, s));
    AID_TRY(philox_fill(u2, B, 1, 0.f, seed, offset + (uint64_t)(B + 3) / 4, s));
    hipLaunchKernelGGL(sample_time_kernel, dim3(1), dim3(256), 0, s, time_importance, first_call, u1, u2, t_out, B);
    AID_LAUNCH_CHECK();
    return AID_OK;
}

int aid_elbo_step(const AidScoreNet* net, const AidScoreNet* grads, const AidDiffusionParams* dp, const AidDiffusionParams* dgrads,
                  const AidDecoderNet* dec, const AidRewardNet* rew, int B, const float* obs, const float* rewards, const float* latents,
                  const AidElboNoise* noise, const AidElboCfg* cfg, float* time_importance, float* t_out, float* per_sample_out,
                  float* metrics, void* wsp, size_t ws_bytes, void* stream) {
    AID_TRY(score_validate(net)); AID_TRY(score_validate(grads));
    AID_REQUIRE(dp && dgrads && rew && obs && rewards && latents && noise && cfg && time_importance && metrics, AID_E_NULL,
                "aid_elbo_step: null argument");
    float* d_obs = noise->d_obs_out;   // optional [B,O]: accumulates d(loss)/d(observations) (pixel mode)
    AID_REQUIRE(B > 0, AID_E_SHAPE, "aid_elbo_step: B=%d", B);
    hipStream_t s = (hipStream_t)stream;
    const AidScoreNet& n = *net;
    const int D = n.D, L = n.L, O = dec ? dec->O : n.O, NA = 2 * n.n_blocks + 1, ldmod = NA * 2 * D;
    const bool train = cfg->train != 0;
    Workspace ws(wsp, ws_bytes);
    SplitScope split(ws, s);           // the 256-row x 512-column products of both passes: K slices folded in the kernel (gemm_mid.hip)
    uint64_t ctr = cfg->offset;
    auto draw = [&](const float* given, size_t cnt, int kind, float p, const float** out) -> int {
        if (given) { *out = given; return AID_OK; }
        float* buf = ws.f32(cnt); AID_WS_OK(ws);
        AID_TRY(philox_fill(buf, cnt, kind, p, cfg->seed, ctr, s)); ctr += (cnt + 3) / 4; *out = buf; return AID_OK;
    };
    // ---- noise (draw order of the reference: decoder masks, t, eps, score mask, GP mask, prior eps)
    const float* dmask[3] = {nullptr, nullptr, nullptr};
    if (train && dec) {
        AID_TRY(draw(noise->dec_mask0, (size_t)B * dec->s[0].out, 2, 0.2f, &dmask[0]));
        AID_TRY(draw(noise->dec_mask1, (size_t)B * dec->s[1].out, 2, 0.2f, &dmask[1]));
        AID_TRY(draw(noise->dec_mask2, (size_t)B * dec->s[2].out, 2, 0.2f, &dmask[2]));
    }
    const float* t = noise->t;
    if (!t) {
        const float *u1, *u2; float* tb = ws.f32(B);
        AID_TRY(draw(nullptr, B, 1, 0.f, &u1)); AID_TRY(draw(nullptr, B, 1, 0.f, &u2)); AID_WS_OK(ws);
        hipLaunchKernelGGL(sample_time_kernel, dim3(1), dim3(256), 0, s, time_importance, cfg->first_call, u1, u2, tb, B);
        AID_LAUNCH_CHECK(); t = tb;
    }
    const float *eps, *smask = nullptr, *gmask = nullptr, *peps;
    AID_TRY(draw(noise->eps, (size_t)B * L, 0, 0.f, &eps));
    if (train) { AID_TRY(draw(noise->score_mask, (size_t)B * D, 2, 0.1f, &smask)); AID_TRY(draw(noise->gp_mask, (size_t)B * D, 2, 0.1f, &gmask)); }
    AID_TRY(draw(noise->prior_eps, (size_t)B * L, 0, 0.f, &peps));

    float* zt = ws.f32((size_t)B * L); float* score = ws.f32((size_t)B * L); float* d_score = ws.f32((size_t)B * L);
    float* dz = ws.f32((size_t)B * L); float* gvec = ws.f32((size_t)B * L); float* wvec = ws.f32((size_t)B * L);
    float* per = ws.f32(B); float* wts = ws.f32(B); float* dlamA = ws.f32(B); float* tmp = ws.f32(8);
    float* dmod = ws.f32((size_t)B * ldmod); float* rec = ws.f32((size_t)B * O);
    AID_WS_OK(ws);

    // ---- reconstruction term (value only; decoder never trained in state mode)
    if (dec) {   // dec == NULL: the caller already stored the reconstruction term in metrics[0] (pixel mode: feature decoder)
        AID_TRY(decoder_fwd(*dec, B, latents, L, train ? dmask : nullptr, rec, O, ws, s));
        AID_TRY(mse_scalar(rec, O, obs, O, B, O, metrics + 0, s));
    }

    // ---- main score-matching pass
    hipLaunchKernelGGL(elbo_prepare_kernel, dim3((B * L + 255) / 256), dim3(256), 0, s, latents, t, eps, dp->log_snr_min, dp->log_snr_max, zt, B, L);
    AID_LAUNCH_CHECK();
    ScoreActs am;
    AID_TRY(score_acts_alloc(n, B, true, ws, am));
    AID_TRY(score_time_embed(n, B, t, AID_T_CONTINUOUS, am, s));
    AID_TRY(score_obs_stage0(n, B, obs, am, s));
    AID_TRY(score_cond(n, B, smask, am.temb, D, am, s));
    AID_TRY(score_trunk(n, B, zt, am, true, s));
    AID_TRY(score_head(am.raw, n.output_multiplier, t, AID_T_CONTINUOUS, score, B, L, s));
    hipLaunchKernelGGL(elbo_score_loss_kernel, dim3((B + 255) / 256), dim3(256), 0, s, score, eps, t, dp->log_snr_min, dp->log_snr_max,
                       cfg->diffusion_weight, per, wts, d_score, dlamA, B, L);
    AID_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_scalar_kernel, dim3(1), dim3(1024), 0, s, per, B, metrics + 2);
    hipLaunchKernelGGL(mean_scalar_kernel, dim3(1), dim3(1024), 0, s, wts, B, metrics + 7);
    AID_LAUNCH_CHECK();
    // every weight / bias gradient of a pass is recorded while its adjoint chain runs and computed by one grouped launch at the end of the pass
    GradQueue gq(s);
    {
        const size_t mk = ws.mark();
        AID_TRY(score_reverse(n, grads, B, am, nullptr, zt, nullptr, d_score, t, AID_T_CONTINUOUS, 0.f, dz, dmod, ws, s, &gq));
        AID_TRY(score_cond_backward(n, *grads, B, am, dmod, obs, smask, t, AID_T_CONTINUOUS, ws, s, d_obs, &gq));
        AID_TRY(gq.flush());
        ws.release(mk);
    }
    hipLaunchKernelGGL(elbo_lambda_grad_kernel, dim3(1), dim3(1024), 0, s, dlamA, dz, latents, eps, t, dp->log_snr_min, dp->log_snr_max,
                       dgrads->log_snr_min, dgrads->log_snr_max, B, L);
    AID_LAUNCH_CHECK();

    // ---- gradient penalty (separate forward with its own dropout mask, active_inference.py:716-727)
    ScoreActs ag; ScoreTan tan;
    AID_TRY(score_acts_alloc(n, B, true, ws, ag));
    AID_TRY(score_tan_alloc(n, B, ws, tan)); AID_WS_OK(ws);
    AID_TRY(score_time_embed(n, B, t, AID_T_CONTINUOUS, ag, s));
    AID_TRY(score_obs_stage0(n, B, obs, ag, s));
    AID_TRY(score_cond(n, B, gmask, ag.temb, D, ag, s));
    AID_TRY(score_trunk(n, B, zt, ag, true, s));
    AID_TRY(score_reverse(n, nullptr, B, ag, nullptr, zt, nullptr, nullptr, t, AID_T_CONTINUOUS, 0.f, gvec, nullptr, ws, s));   // g = d(sum s)/dz_t
    AID_TRY(gp_weights(gvec, wvec, metrics + 5, B, L, s));
    AID_TRY(score_tangent_forward(n, B, wvec, ag, tan, s));
    {
        const size_t mk = ws.mark();
        AID_TRY(score_reverse(n, grads, B, ag, &tan, zt, wvec, nullptr, t, AID_T_CONTINUOUS, -cfg->gp_weight, nullptr, dmod, ws, s, &gq));
        AID_TRY(score_cond_backward(n, *grads, B, ag, dmod, obs, gmask, t, AID_T_CONTINUOUS, ws, s, d_obs, &gq));
        AID_TRY(gq.flush());
        ws.release(mk);
    }

    // ---- prior / KL term, reward NLL, metrics, time-importance EMA
    hipLaunchKernelGGL(kl_prior_kernel, dim3(1), dim3(1024), 0, s, latents, peps, dp->prior_mean, dp->prior_log_std, t, cfg->kl_weight,
                       dgrads->prior_mean, dgrads->prior_log_std, tmp, B, L, noise->mean_t);
    AID_LAUNCH_CHECK();
    {
        RewardSave rs; const size_t mk = ws.mark();
        AID_TRY(reward_fwd(*rew, B, latents, L, rs, ws, s));
        hipLaunchKernelGGL(reward_nll_kernel, dim3(1), dim3(1024), 0, s, rs.out, rewards, B, metrics + 4);
        AID_LAUNCH_CHECK();
        ws.release(mk);
    }
    hipLaunchKernelGGL(elbo_metrics_kernel, dim3(1), dim3(64), 0, s, metrics, tmp, cfg->kl_weight, cfg->diffusion_weight, cfg->reward_weight, cfg->gp_weight);
    if (!cfg->defer_time_importance) {
        hipLaunchKernelGGL(time_importance_update_kernel, dim3(1), dim3(128), 0, s, time_importance, t, per, B, cfg->first_call);
        AID_LAUNCH_CHECK();
    }
    if (t_out) AID_TRY(copy_f32(t, t_out, B, s));
    if (per_sample_out) AID_TRY(copy_f32(per, per_sample_out, B, s));
    return AID_OK;
}

}  // extern "C"
