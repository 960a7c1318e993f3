// lambda-returns, value / dynamics optimiser phases, forward-only helpers (act()), replay gather, Fokker-Planck.
// Reference: aid/core/active_inference.py:638-707; aid/agents/state_agent.py:183-241; aid/utils/buffers.py:83-111;
// aid/core/belief_dynamics.py:97-172,368-389.
#include "mlp.h"

namespace aid {

// One thread per minibatch position; float32 accumulation order mirrors the reference's tensor arithmetic
// (python-float scalars are rounded to float32 when they meet a tensor).  O(B * n^2), n <= 5.
__global__ void lambda_returns_kernel(const float* __restrict__ r, const float* __restrict__ nv, const float* __restrict__ done,
                                      int B, float gamma, double lam, int n_steps, int excl, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B) return;
    float rets[16];
    int cnt = 0;
    const int nmax = min(n_steps + 1, B - idx);
    for (int n = 1; n < nmax && cnt < 16; ++n) {
        float ret = 0.f, disc = 1.0f;
        for (int k = 0; k < n; ++k) {
            if (!(excl && k == 0)) ret = ret + disc * r[idx + k];       // exclude_immediate_rewards (:674-675)
            const float nd = done[idx + k] != 0.f ? 0.f : gamma;
            disc = disc * nd;
        }
        if (idx + n < B && done[idx + n - 1] == 0.f) ret = ret + disc * nv[idx + n];
        rets[cnt++] = ret;
    }
    if (cnt > 0) {
        float wr = 0.f; double ls = 0.0, li = 1.0;
        for (int i = 0; i < cnt - 1; ++i) { const double w = (1.0 - lam) * li; wr = wr + (float)w * rets[i]; ls += w; li *= lam; }
        double lw = 1.0; for (int i = 0; i < cnt - 1; ++i) lw *= lam;
        wr = wr + (float)lw * rets[cnt - 1]; ls += lw;
        out[idx] = wr / (float)(ls + 1e-8);
    } else {
        const float boot = (done[idx] != 0.f ? 0.f : gamma) * nv[idx];
        out[idx] = excl ? boot : r[idx] + boot;
    }
}

__device__ __forceinline__ float wgs(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < 64) { t = (threadIdx.x < (blockDim.x >> 6)) ? sh[threadIdx.x] : 0.f; t = wave_sum(t); }
    return t;
}
// F.huber_loss(values, targets) (delta=1, mean) and its gradient wrt values
__global__ void huber_kernel(const float* __restrict__ v, const float* __restrict__ tg, int B, float* __restrict__ dv, float* __restrict__ loss) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float d = v[i] - tg[i], a = fabsf(d);
        acc += a < 1.f ? 0.5f * d * d : a - 0.5f;
        dv[i] = fminf(fmaxf(d, -1.f), 1.f) / (float)B;
    }
    const float s = wgs(acc, sh);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}
__global__ void mse_grad_kernel(const float* __restrict__ p, const float* __restrict__ tg, size_t n, float* __restrict__ dp, float* __restrict__ loss) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) { const float d = p[i] - tg[i]; acc += d * d; dp[i] = 2.f * d / (float)n; }
    const float s = wgs(acc, sh);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}

__global__ void replay_gather_kernel(const float* __restrict__ obs, const float* __restrict__ nobs, const float* __restrict__ act,
                                     const float* __restrict__ rew, const float* __restrict__ done, const int64_t* __restrict__ idx,
                                     int B, int O, int A, float* __restrict__ o_out, float* __restrict__ n_out, float* __restrict__ a_out,
                                     float* __restrict__ r_out, float* __restrict__ d_out) {
    const int W = 2 * O + A + 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * W) return;
    const int b = i / W, c = i % W;
    const int64_t row = idx[b];
    if (c < O) o_out[(size_t)b * O + c] = obs[row * O + c];
    else if (c < 2 * O) n_out[(size_t)b * O + (c - O)] = nobs[row * O + (c - O)];
    else if (c < 2 * O + A) a_out[(size_t)b * A + (c - 2 * O)] = act[row * A + (c - 2 * O)];
    else if (c == 2 * O + A) r_out[b] = rew[row];
    else d_out[b] = done[row];
}

// uint8 frame storage (pixel replay, 4x less HBM): obs / next_obs rows are bytes, widened to float with the same value (0..255)
__global__ void replay_gather_u8_kernel(const uint8_t* __restrict__ obs, const uint8_t* __restrict__ nobs, const float* __restrict__ act,
                                        const float* __restrict__ rew, const float* __restrict__ done, const int64_t* __restrict__ idx,
                                        int B, int O, int A, float* __restrict__ o_out, float* __restrict__ n_out, float* __restrict__ a_out,
                                        float* __restrict__ r_out, float* __restrict__ d_out) {
    const size_t W = (size_t)2 * O + A + 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * W) return;
    const int b = i / W; const size_t c = i % W;
    const int64_t row = idx[b];
    if (c < (size_t)O) o_out[(size_t)b * O + c] = (float)obs[row * O + c];
    else if (c < (size_t)2 * O) n_out[(size_t)b * O + (c - O)] = (float)nobs[row * O + (c - O)];
    else if (c < (size_t)2 * O + A) a_out[(size_t)b * A + (c - 2 * O)] = act[row * A + (c - 2 * O)];
    else if (c == (size_t)2 * O + A) r_out[b] = rew[row];
    else d_out[b] = done[row];
}

// fp64 diagonal Fokker-Planck belief step (closed form; see oracle.aid_oracle.fokker_planck_step)
__global__ void fokker_planck_kernel(double* __restrict__ mean, double* __restrict__ var, const double* __restrict__ obs,
                                     const double* __restrict__ score, const double* __restrict__ xi, int n, int L, double dt, double D,
                                     double lr, double ns, double min_var, double max_var, double* __restrict__ ent) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const double ns2 = ns * ns;
    double g2 = 0.0;
    for (int l = 0; l < L; ++l) {
        const double m = mean[(size_t)b * L + l];
        const double g = -(m - obs[(size_t)b * L + l]) / ns2 - m + score[(size_t)b * L + l];
        g2 += g * g;
    }
    const double adt = dt / (1.0 + 0.1 * sqrt(g2));
    const double nz = sqrt(2.0 * D * dt) * ns;
    const double h = -(1.0 / ns2 + 1.0);
    const double fac = exp((-2.0 * h + 2.0 * D) * dt);
    const double lo = fmax(min_var, 1e-8);
    double e = 0.0;
    for (int l = 0; l < L; ++l) {
        const size_t i = (size_t)b * L + l;
        const double m = mean[i];
        const double g = -(m - obs[i]) / ns2 - m + score[i];
        mean[i] = m + (-lr * g) * adt + nz * xi[i];
        const double v = fmin(fmax(var[i] * fac, lo), max_var);
        var[i] = v;
        e += 0.5 * (log(2.0 * 3.14159265358979323846 * 2.718281828459045) + log(v));
    }
    if (ent) ent[b] = e;
}

// ---- full-covariance Fokker-Planck (belief_dynamics.py:141-152, 268-342) --------------------------------------------------------------
// With the Gaussian observation model the Hessian is h I, so E = expm((-H - H^T + 2 D I) dt) = e^{(-2h+2D) dt} I and every later operation
// (eigen-clamp, condition-number regularisation, (cov + eps I)^-1, logdet) is a spectral function of the covariance: its eigenvectors never
// change.  The state is therefore kept as (lam, V) — one symmetric eigen-decomposition at reset (sym_eig_jacobi_kernel), then per step a
// diagonal update of lam and the two rebuilds  cov = V diag(lam) V^T,  precision = V diag(1 / (lam + eps)) V^T.

// Parallel cyclic Jacobi, fp64, one workgroup: A [n,n] symmetric (destroyed) -> eigenvalues w [n] (unsorted) and eigenvectors V (columns).
// Each round rotates n/2 disjoint (p, q) pairs of a round-robin tournament: rows first (A <- J^T A), then columns (A <- A J, V <- V J).
__global__ void __launch_bounds__(256) sym_eig_jacobi_kernel(double* __restrict__ A, double* __restrict__ w, double* __restrict__ V, int n) {
    __shared__ double cs[128][2];
    __shared__ int pq[128][2];
    __shared__ double red[256];
    __shared__ int done;
    const int tid = threadIdx.x, NT = blockDim.x;
    for (int i = tid; i < n * n; i += NT) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    const int m = n + (n & 1), half = m / 2;
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = tid; i < n * n; i += NT) { const double a = A[i]; if (i / n == i % n) dg += a * a; else off += a * a; }
        red[tid] = off; __syncthreads();
        for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
        off = red[0]; __syncthreads();
        red[tid] = dg; __syncthreads();
        for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
        dg = red[0];
        if (tid == 0) done = !(off > 1e-30 * dg) || !(off == off);      // relative off-diagonal mass below 1e-15^2, or NaN input
        __syncthreads();
        if (done) break;
        for (int round = 0; round < m - 1; ++round) {
            if (tid < half) {
                const int a = tid == 0 ? m - 1 : (round + tid) % (m - 1);
                const int b = (round + m - 1 - tid) % (m - 1);
                const int p = min(a, b), q = max(a, b);
                double c = 1.0, sn = 0.0;
                if (q < n) {
                    const double apq = A[p * n + q];
                    if (apq != 0.0) {
                        const double tau = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t); sn = t * c;
                    }
                }
                cs[tid][0] = c; cs[tid][1] = sn; pq[tid][0] = p; pq[tid][1] = q;
            }
            __syncthreads();
            for (int e = tid; e < half * n; e += NT) {            // rows p, q of A
                const int k = e / n, j = e % n, p = pq[k][0], q = pq[k][1];
                if (q < n) {
                    const double c = cs[k][0], sn = cs[k][1], x = A[p * n + j], y = A[q * n + j];
                    A[p * n + j] = c * x - sn * y; A[q * n + j] = sn * x + c * y;
                }
            }
            __syncthreads();
            for (int e = tid; e < half * n; e += NT) {            // columns p, q of A and of V
                const int k = e / n, i = e % n, p = pq[k][0], q = pq[k][1];
                if (q < n) {
                    const double c = cs[k][0], sn = cs[k][1];
                    double x = A[i * n + p], y = A[i * n + q];
                    A[i * n + p] = c * x - sn * y; A[i * n + q] = sn * x + c * y;
                    x = V[i * n + p]; y = V[i * n + q];
                    V[i * n + p] = c * x - sn * y; V[i * n + q] = sn * x + c * y;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += NT) w[i] = A[i * n + i];
}

// One workgroup per belief.  advance = 1: mean step (as the diagonal kernel) and lam *= e^{2 (-2h+2D) dt};  advance = 0: only the
// stabilisation of `reset(initial_cov=...)`.  Then clamp / regularise lam (:296-332) and rebuild cov, precision (:334-342), entropy (:368-383).
__global__ void __launch_bounds__(256) fokker_planck_full_kernel(double* __restrict__ mean, double* __restrict__ lam, const double* __restrict__ V,
                                                                 const double* __restrict__ obs, const double* __restrict__ score,
                                                                 const double* __restrict__ xi, int L, double dt, double D, double lr, double ns,
                                                                 double min_eig, int advance, double* __restrict__ cov, double* __restrict__ prec,
                                                                 double* __restrict__ ent, double* __restrict__ cond_out) {
    __shared__ double red[256];
    __shared__ double sl[128], sinv[128];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    mean += (size_t)b * L; lam += (size_t)b * L; V += (size_t)b * L * L; cov += (size_t)b * L * L; prec += (size_t)b * L * L;
    const double ns2 = ns * ns;
    if (advance) {
        obs += (size_t)b * L; score += (size_t)b * L; xi += (size_t)b * L;
        double g = 0.0;
        if (tid < L) { const double m = mean[tid]; g = -(m - obs[tid]) / ns2 - m + score[tid]; }
        red[tid] = g * g; __syncthreads();
        for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
        const double adt = dt / (1.0 + 0.1 * sqrt(red[0]));
        __syncthreads();
        if (tid < L) mean[tid] = mean[tid] + (-lr * g) * adt + sqrt(2.0 * D * dt) * ns * xi[tid];
    }
    const double h = -(1.0 / ns2 + 1.0);
    const double fac = advance ? exp(2.0 * (-2.0 * h + 2.0 * D) * dt) : 1.0;
    double l = tid < L ? fmax(lam[tid] * fac, min_eig) : 0.0;
    // max, min, mean of the clamped eigenvalues
    red[tid] = tid < L ? l : -1.0; __syncthreads();
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] = fmax(red[tid], red[tid + st]); __syncthreads(); }
    const double lmax = red[0]; __syncthreads();
    red[tid] = tid < L ? l : 1.7976931348623157e308; __syncthreads();
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] = fmin(red[tid], red[tid + st]); __syncthreads(); }
    const double lmin = red[0]; __syncthreads();
    red[tid] = l; __syncthreads();
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double lmean = red[0] / L; __syncthreads();
    const double cond = lmax / lmin;
    if (cond > 1e6) l += lmean * 1e-6;
    if (tid < L) { lam[tid] = l; sl[tid] = l; sinv[tid] = 1.0 / (l + min_eig); }
    red[tid] = tid < L ? log(l) : 0.0; __syncthreads();
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    if (tid == 0) {
        if (ent) ent[b] = 0.5 * (L * log(2.0 * 3.14159265358979323846 * 2.718281828459045) + red[0]);
        if (cond_out) cond_out[b] = cond;
    }
    for (int e = tid; e < L * L; e += NT) {
        const int i = e / L, j = e % L;
        double c = 0.0, p = 0.0;
        for (int k = 0; k < L; ++k) { const double vv = V[i * L + k] * V[j * L + k]; c += vv * sl[k]; p += vv * sinv[k]; }
        cov[e] = c; prec[e] = p;
    }
}

// p_sample with a given score (diffusion.py:208-255): z0 = (z + c0 s) c1 ; mean = c2 z0 + c3 z ; + c4 eps for t > 0
__global__ void p_sample_kernel(const float* __restrict__ z, const float* __restrict__ score, const float* __restrict__ eps,
                                const float* __restrict__ coef, int step, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float zt = z[i];
    const float z0 = (zt + coef[0] * score[i]) * coef[1];
    float v = coef[2] * z0 + coef[3] * zt;
    if (step > 0 && eps) v += coef[4] * eps[i];
    out[i] = v;
}
// RunningMeanStd.update + normalize (aid/agents/base_agent.py:24-52) in fp64 on device, no host round trip.
// state: [3] doubles (mean, var, count).
__global__ void reward_normalize_kernel(const float* __restrict__ r, int B, double* __restrict__ state, float* __restrict__ out,
                                        const float* __restrict__ loc, int Bloc) {
    __shared__ double sh[16];
    __shared__ double s_mean, s_var;
    double acc = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) acc += (double)r[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w]; s_mean = t / B; }
    __syncthreads();
    const double bm = s_mean;
    acc = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { const double d = (double)r[i] - bm; acc += d * d; }
    acc = wave_sum_d(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
        const double bv = t / B, bc = (double)B;
        const double mean = state[0], var = state[1], cnt = state[2];
        const double delta = bm - mean, tot = cnt + bc;
        const double new_mean = mean + delta * bc / tot;
        const double m2 = var * cnt + bv * bc + delta * delta * cnt * bc / tot;
        state[0] = new_mean; state[1] = m2 / tot; state[2] = tot;
        s_mean = new_mean; s_var = m2 / tot;
    }
    __syncthreads();
    const double mu = s_mean, sd = sqrt(s_var + 1e-8);
    for (int i = threadIdx.x; i < Bloc; i += blockDim.x) out[i] = (float)(((double)loc[i] - mu) / sd);
}

// FreeEnergyComputation.compute_loss reductions (free_energy.py:62-78): out = mean_b sum_l of
// (z-mu_p)^2, (o-z)^2, s^2
__global__ void free_energy_terms_kernel(const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ o,
                                         const float* __restrict__ sc, int B, int L, float* __restrict__ out) {
    __shared__ float sh[16];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const float d0 = z[i] - mu[i], d1 = o[i] - z[i], s_ = sc[i];
        a0 += d0 * d0; a1 += d1 * d1; a2 += s_ * s_;
    }
    const float s0 = wgs(a0, sh); const float s1 = wgs(a1, sh); const float s2 = wgs(a2, sh);
    if (threadIdx.x == 0) { out[0] = s0 / (float)B; out[1] = s1 / (float)B; out[2] = s2 / (float)B; }
}

}  // namespace aid

using namespace aid;

extern "C" {

int aid_lambda_returns_ex(const float* rewards, const float* next_values, const float* dones, int B, float gamma, double lam,
                          int n_steps, int exclude_immediate_rewards, float* out, void* stream) {
    AID_REQUIRE(rewards && next_values && dones && out, AID_E_NULL, "aid_lambda_returns: null tensor");
    AID_REQUIRE(B > 0 && n_steps >= 1 && n_steps <= 16, AID_E_SHAPE, "aid_lambda_returns: B=%d n_steps=%d", B, n_steps);
    hipLaunchKernelGGL(lambda_returns_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, rewards, next_values, dones, B, gamma,
                       lam, n_steps, exclude_immediate_rewards ? 1 : 0, out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_lambda_returns(const float* rewards, const float* next_values, const float* dones, int B, float gamma, float lam,
                       int n_steps, float* out, void* stream) {
    return aid_lambda_returns_ex(rewards, next_values, dones, B, gamma, (double)lam, n_steps, 0, out, stream);
}

size_t aid_small_net_workspace(int B, int D) { return sizeof(float) * (size_t)B * (64 * (size_t)D + 1024) + (1 << 16); }
size_t aid_value_step_workspace(const AidValueNet* val, int B) { return val ? 2 * aid_small_net_workspace(B + 64, val->D) + sizeof(float) * (size_t)(B + 64) * 4 * val->D : 0; }
size_t aid_dynamics_step_workspace(const AidDynamicsNet* dyn, int B) { return dyn ? aid_small_net_workspace(B, dyn->D) + sizeof(float) * (size_t)(B + 64) * 4 * dyn->D : 0; }

int aid_value_step(const AidValueNet* val, const AidValueNet* vgrads, int B, const float* latents, const float* next_latents,
                   const float* rewards, const float* dones, float gamma, double lam, int n_steps, float* targets_out, float* loss_out,
                   void* wsp, size_t ws_bytes, void* stream) {
    return aid_value_step_ex(val, vgrads, B, B, latents, next_latents, rewards, dones, gamma, lam, n_steps, targets_out, loss_out, wsp, ws_bytes, stream);
}

int aid_value_step_ex(const AidValueNet* val, const AidValueNet* vgrads, int B, int Bx, const float* latents, const float* next_latents,
                      const float* rewards, const float* dones, float gamma, double lam, int n_steps, float* targets_out, float* loss_out,
                      void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(val && vgrads && latents && next_latents && rewards && dones && loss_out, AID_E_NULL, "aid_value_step: null argument");
    AID_REQUIRE(B > 0 && Bx >= B && Bx <= B + 64, AID_E_SHAPE, "aid_value_step: B=%d B_ext=%d", B, Bx);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    float* t0 = ws.f32(B); float* t1 = ws.f32(Bx); float* v = ws.f32(B); float* nv = ws.f32(Bx); float* tg = ws.f32(Bx); float* dv = ws.f32(B);
    AID_WS_OK(ws);
    AID_TRY(fill_const(t0, 0.f, B, s)); AID_TRY(fill_const(t1, 1.f, Bx, s));
    ValueSave sv, sv2;
    AID_TRY(value_fwd(*val, B, latents, val->L, t0, v, sv, ws, s));
    { const size_t mk = ws.mark(); AID_TRY(value_fwd(*val, Bx, next_latents, val->L, t1, nv, sv2, ws, s)); ws.release(mk); }
    AID_TRY(aid_lambda_returns_ex(rewards, nv, dones, Bx, gamma, lam, n_steps, 0, tg, stream));   // rows >= B: halo, their targets are unused
    hipLaunchKernelGGL(huber_kernel, dim3(1), dim3(1024), 0, s, v, tg, B, dv, loss_out);
    AID_LAUNCH_CHECK();
    GradQueue gq(s);          // the five Linear weight / bias gradients in one grouped launch (this step sits at the very end of the train_step's critical path)
    AID_TRY(value_bwd(*val, vgrads, B, t0, sv, dv, nullptr, 0, ws, s, &gq));
    AID_TRY(gq.flush());
    if (targets_out) AID_TRY(copy_f32(tg, targets_out, B, s));
    return AID_OK;
}

int aid_dynamics_step(const AidDynamicsNet* dyn, const AidDynamicsNet* dgrads, int B, const float* latents, const float* actions,
                      const float* next_latents, float* loss_out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(dyn && dgrads && latents && actions && next_latents && loss_out, AID_E_NULL, "aid_dynamics_step: null argument");
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes);
    const int L = dyn->L;
    float* pred = ws.f32((size_t)B * L); float* dp = ws.f32((size_t)B * L);
    DynSave sv;
    AID_TRY(dynamics_fwd(*dyn, B, latents, L, actions, dyn->A, pred, sv, ws, s));
    hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(1024), 0, s, pred, next_latents, (size_t)B * L, dp, loss_out);
    AID_LAUNCH_CHECK();
    GradQueue gq(s);
    AID_TRY(dynamics_bwd(*dyn, dgrads, B, sv, dp, nullptr, nullptr, ws, s, &gq));
    AID_TRY(gq.flush());
    return AID_OK;
}

int aid_policy_forward(const AidPolicyNet* pol, int B, const float* z, const float* eps, float* action, float* log_prob, float* entropy,
                       void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(pol && z && action, AID_E_NULL, "aid_policy_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(wsp, ws_bytes); PolicySave sv;
    AID_TRY(policy_fwd(*pol, B, z, pol->L, eps, sv, ws, s));
    AID_TRY(copy_f32(sv.action, action, (size_t)B * pol->A, s));
    if (log_prob) AID_TRY(copy_f32(sv.logp, log_prob, B, s));
    if (entropy) AID_TRY(copy_f32(sv.ent, entropy, B, s));
    return AID_OK;
}
int aid_value_forward(const AidValueNet* val, int B, const float* state, const float* t, float* out, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(val && state && t && out, AID_E_NULL, "aid_value_forward: null argument");
    Workspace ws(wsp, ws_bytes); ValueSave sv;
    return value_fwd(*val, B, state, val->L, t, out, sv, ws, (hipStream_t)stream);
}
int aid_predict_next_latent(const AidDynamicsNet* dyn, int B, const float* z, const float* a, float* next_mean, void* wsp, size_t ws_bytes,
                            void* stream) {
    AID_REQUIRE(dyn && z && a && next_mean, AID_E_NULL, "aid_predict_next_latent: null argument");
    Workspace ws(wsp, ws_bytes); DynSave sv;
    return dynamics_fwd(*dyn, B, z, dyn->L, a, dyn->A, next_mean, sv, ws, (hipStream_t)stream);
}
int aid_reward_forward(const AidRewardNet* rew, int B, const float* z, float* out2, void* wsp, size_t ws_bytes, void* stream) {
    AID_REQUIRE(rew && z && out2, AID_E_NULL, "aid_reward_forward: null argument");
    Workspace ws(wsp, ws_bytes); RewardSave sv;
    AID_TRY(reward_fwd(*rew, B, z, rew->L, sv, ws, (hipStream_t)stream));
    return copy_f32(sv.out, out2, (size_t)B * 2, (hipStream_t)stream);
}

int aid_p_sample(const float* z, const float* score, const float* eps, const float* coefs, int step, int B, int L, float* out, void* stream) {
    AID_REQUIRE(z && score && coefs && out, AID_E_NULL, "aid_p_sample: null tensor");
    const int n = B * L;
    if (n == 0) return AID_OK;
    hipLaunchKernelGGL(p_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, z, score, eps, coefs + 5 * step, step, out, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}
int aid_reward_normalize(const float* rewards, int B, double* state, float* out, void* stream) {
    AID_REQUIRE(rewards && state && out, AID_E_NULL, "aid_reward_normalize: null tensor");
    AID_REQUIRE(B > 0, AID_E_SHAPE, "aid_reward_normalize: B=%d", B);
    hipLaunchKernelGGL(reward_normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rewards, B, state, out, rewards, B);
    AID_LAUNCH_CHECK(); return AID_OK;
}
int aid_reward_normalize_global(const float* stats_src, int n_stats, const float* rewards, int B, double* state, float* out, void* stream) {
    AID_REQUIRE(stats_src && rewards && state && out, AID_E_NULL, "aid_reward_normalize_global: null tensor");
    AID_REQUIRE(B > 0 && n_stats > 0, AID_E_SHAPE, "aid_reward_normalize_global: B=%d n=%d", B, n_stats);
    hipLaunchKernelGGL(reward_normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stats_src, n_stats, state, out, rewards, B);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_free_energy_terms(const float* states, const float* prior_mean, const float* obs, const float* score, int B, int L, float* out3,
                          void* stream) {
    AID_REQUIRE(states && prior_mean && obs && score && out3, AID_E_NULL, "aid_free_energy_terms: null tensor");
    hipLaunchKernelGGL(free_energy_terms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, states, prior_mean, obs, score, B, L, out3);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_replay_gather(const float* obs, const float* next_obs, const float* actions, const float* rewards, const float* dones,
                      const int64_t* idx, int B, int O, int A, float* obs_out, float* next_obs_out, float* act_out, float* rew_out,
                      float* done_out, void* stream) {
    AID_REQUIRE(obs && next_obs && actions && rewards && dones && idx && obs_out && next_obs_out && act_out && rew_out && done_out, AID_E_NULL,
                "aid_replay_gather: null tensor");
    const int n = B * (2 * O + A + 2);
    if (n == 0) return AID_OK;
    hipLaunchKernelGGL(replay_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs, next_obs, actions, rewards, dones,
                       idx, B, O, A, obs_out, next_obs_out, act_out, rew_out, done_out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_replay_gather_u8(const uint8_t* obs, const uint8_t* next_obs, const float* actions, const float* rewards, const float* dones,
                         const int64_t* idx, int B, int O, int A, float* obs_out, float* next_obs_out, float* act_out, float* rew_out,
                         float* done_out, void* stream) {
    AID_REQUIRE(obs && next_obs && actions && rewards && dones && idx && obs_out && next_obs_out && act_out && rew_out && done_out, AID_E_NULL,
                "aid_replay_gather_u8: null tensor");
    const size_t n = (size_t)B * ((size_t)2 * O + A + 2);
    if (n == 0) return AID_OK;
    hipLaunchKernelGGL(replay_gather_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, next_obs, actions, rewards,
                       dones, idx, B, O, A, obs_out, next_obs_out, act_out, rew_out, done_out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_fokker_planck_step(double* mean, double* var, const double* obs, const double* score, const double* xi, int n, int L, double dt,
                           double D, double lr, double noise_scale, double min_var, double max_var, double* entropy_out, void* stream) {
    AID_REQUIRE(mean && var && obs && score && xi, AID_E_NULL, "aid_fokker_planck_step: null tensor");
    AID_REQUIRE(n > 0 && L > 0, AID_E_SHAPE, "aid_fokker_planck_step: n=%d L=%d", n, L);
    hipLaunchKernelGGL(fokker_planck_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, mean, var, obs, score, xi, n, L, dt, D, lr,
                       noise_scale, min_var, max_var, entropy_out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_sym_eig_f64(double* A, double* w, double* V, int n, void* stream) {
    AID_REQUIRE(A && w && V, AID_E_NULL, "aid_sym_eig_f64: null tensor");
    AID_REQUIRE(n >= 1 && n <= 256, AID_E_SHAPE, "aid_sym_eig_f64: n=%d (1..256)", n);
    hipLaunchKernelGGL(sym_eig_jacobi_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A, w, V, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int aid_fokker_planck_full_step(double* mean, double* lam, const double* V, const double* obs, const double* score, const double* xi, int n, int L,
                                double dt, double D, double lr, double noise_scale, double min_var, int advance, double* cov_out,
                                double* prec_out, double* entropy_out, double* cond_out, void* stream) {
    AID_REQUIRE(mean && lam && V && cov_out && prec_out, AID_E_NULL, "aid_fokker_planck_full_step: null tensor");
    AID_REQUIRE(!advance || (obs && score && xi), AID_E_NULL, "aid_fokker_planck_full_step: advance needs obs, score, xi");
    AID_REQUIRE(n > 0 && L > 0 && L <= 128, AID_E_SHAPE, "aid_fokker_planck_full_step: n=%d L=%d (L <= 128)", n, L);
    hipLaunchKernelGGL(fokker_planck_full_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, mean, lam, V, obs, score, xi, L, dt, D, lr, noise_scale,
                       fmax(min_var, 1e-8), advance, cov_out, prec_out, entropy_out, cond_out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

}  // extern "C"
