#include "rowwise_bwd.h"

namespace aid {

constexpr int RPB = 4;

// ---------------------------------------------------------------------------------------------- colsum
// Column sums over M rows.  Two deterministic stages so that tall inputs (M = K*B = 2560 rows in the EFE backward) use the
// whole chip instead of N/64 workgroups: stage 1 reduces 256-row slabs into a [slabs][N] partial buffer (a fixed piece of the
// caller's workspace), stage 2 folds the slabs in order.  No atomics: bias gradients are bitwise reproducible.
constexpr int SLAB = 256;
// 1024 threads = 16 waves per 64-column block: a thread walks its rows four at a time with four independent accumulators, so 4 loads are in
// flight per thread and a 256-row slab is 4 dependent steps instead of 64 (these kernels are pure latency at the 256 .. 2 560 rows of the
// ELBO / BPTT backward passes: 10 - 18 us each before, ~70 launches per train_step).  Fixed order: wave partials are folded 0 .. 15.
constexpr int CS_WAVES = 16;
__global__ void __launch_bounds__(64 * CS_WAVES) colsum_slab_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                   const float* __restrict__ stats, float* __restrict__ part0, float* __restrict__ part1, int M, int N) {
    // part0[slab][c] = sum_r a[r,c] * (b ? xhat(b[r,c]) : 1) ; part1[slab][c] = sum_r a[r,c]   (part1 only with b)
    __shared__ float r0[CS_WAVES][64], r1[CS_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, slab = blockIdx.y;
    const int rbeg = slab * SLAB, rend = min(M, rbeg + SLAB);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) for (int r = rbeg + w; r < rend; r += 4 * CS_WAVES) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * CS_WAVES;
            if (rr < rend) {
                const float e = a[(size_t)rr * lda + c];
                if (b) { s0[u] += e * (b[(size_t)rr * ldb + c] - stats[2 * rr]) * stats[2 * rr + 1]; s1[u] += e; }
                else s0[u] += e;
            }
        }
    }
    r0[w][lane] = (s0[0] + s0[1]) + (s0[2] + s0[3]); r1[w][lane] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    __syncthreads();
    if (w == 0 && c < N) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < CS_WAVES; ++i) { t0 += r0[i][lane]; t1 += r1[i][lane]; }
        part0[(size_t)slab * N + c] = t0;
        if (b) part1[(size_t)slab * N + c] = t1;
    }
}
__global__ void colsum_fold_kernel(const float* __restrict__ part, int slabs, float* __restrict__ out, int N, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int i = 0; i < slabs; ++i) s += part[(size_t)i * N + c];
    out[c] = accumulate ? out[c] + s : s;
}
__global__ void __launch_bounds__(64 * CS_WAVES) colsum_kernel(const float* __restrict__ a, int lda, float* __restrict__ out, int M, int N, int accumulate) {
    __shared__ float red[CS_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) for (int r = w; r < M; r += 4 * CS_WAVES) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int rr = r + u * CS_WAVES; if (rr < M) s[u] += a[(size_t)rr * lda + c]; }
    }
    red[w][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (w == 0 && c < N) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < CS_WAVES; ++i) v += red[i][lane];
        out[c] = accumulate ? out[c] + v : v;
    }
}
int colsum(const float* a, int lda, float* out, int M, int N, int accumulate, hipStream_t s, Workspace* ws) {
    if (ws && M > 2 * SLAB) {
        const int slabs = (M + SLAB - 1) / SLAB;
        const size_t mark = ws->mark();
        float* part = ws->f32((size_t)slabs * N);
        if (!ws->overflow) {
            hipLaunchKernelGGL(colsum_slab_kernel, dim3((N + 63) / 64, slabs), dim3(64 * CS_WAVES), 0, s, a, lda, (const float*)nullptr, 0,
                               (const float*)nullptr, part, (float*)nullptr, M, N);
            hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, slabs, out, N, accumulate);
            AID_LAUNCH_CHECK(); ws->release(mark); return AID_OK;
        }
        ws->overflow = false; ws->release(mark);
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(64 * CS_WAVES), 0, s, a, lda, out, M, N, accumulate);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// ---------------------------------------------------------------------------------------------- LN backward
__global__ void ln_bwd_rows_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                   const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   int act, const float* __restrict__ mask, float* __restrict__ dx, int lddx,
                                   float* __restrict__ dyeff, int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * RPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    const float* xr = x + (size_t)row * ldx; const float* dyr = dy + (size_t)row * lddy;
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xh = (xr[c] - mean) * r;
        float e = dyr[c];
        if (mask) e *= mask[(size_t)row * D + c];
        if (act != ACT_NONE) e *= act_grad(gamma ? xh * gamma[c] + beta[c] : xh, act);
        const float g = gamma ? e * gamma[c] : e;
        sg += g; sgx += g * xh;
    }
    sg = wave_sum(sg) / (float)D; sgx = wave_sum(sgx) / (float)D;
    for (int c = lane; c < D; c += 64) {
        const float xh = (xr[c] - mean) * r;
        float e = dyr[c];
        if (mask) e *= mask[(size_t)row * D + c];
        if (act != ACT_NONE) e *= act_grad(gamma ? xh * gamma[c] + beta[c] : xh, act);
        if (dyeff) dyeff[(size_t)row * D + c] = e;
        const float g = gamma ? e * gamma[c] : e;
        if (dx) dx[(size_t)row * lddx + c] = r * (g - sg - xh * sgx);
    }
}
// D = 64 NE known at compile time: every load of the row is issued up front (NE independent loads per operand instead of NE dependent
// round trips) and x-hat / the effective gradient stay in registers between the two passes (no second read, no second act_grad).
// Same per-lane accumulation order as the generic kernel: bit-identical results.
template <int NE, int ACT>
__global__ void __launch_bounds__(256) ln_bwd_rows_reg_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                   const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mask, float* __restrict__ dx, int lddx, float* __restrict__ dyeff, int M) {
    constexpr int D = 64 * NE;
    const int lane = threadIdx.x & 63, row = blockIdx.x * RPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    const float* xr = x + (size_t)row * ldx; const float* dyr = dy + (size_t)row * lddy;
    float xh[NE], e[NE], g[NE], gm[NE], bt[NE], mk[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        xh[i] = xr[c]; e[i] = dyr[c];
        gm[i] = gamma ? gamma[c] : 1.f; bt[i] = gamma ? beta[c] : 0.f;
        mk[i] = mask ? mask[(size_t)row * D + c] : 1.f;
    }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        xh[i] = (xh[i] - mean) * r;
        if (mask) e[i] *= mk[i];
        if (ACT != ACT_NONE) e[i] *= act_grad(gamma ? xh[i] * gm[i] + bt[i] : xh[i], ACT);
        g[i] = gamma ? e[i] * gm[i] : e[i];
        sg += g[i]; sgx += g[i] * xh[i];
    }
    sg = wave_sum(sg) / (float)D; sgx = wave_sum(sgx) / (float)D;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        if (dyeff) dyeff[(size_t)row * D + c] = e[i];
        if (dx) dx[(size_t)row * lddx + c] = r * (g[i] - sg - xh[i] * sgx);
    }
}
template <int NE>
static void launch_ln_bwd_reg(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma, const float* beta, int act,
                              const float* mask, float* dx, int lddx, float* dyeff, int M, hipStream_t s) {
    const dim3 grid((M + RPB - 1) / RPB);
    switch (act) {
        case ACT_SILU: hipLaunchKernelGGL((ln_bwd_rows_reg_kernel<NE, ACT_SILU>), grid, dim3(256), 0, s, dy, lddy, x, ldx, stats, gamma, beta, mask, dx, lddx, dyeff, M); break;
        case ACT_RELU: hipLaunchKernelGGL((ln_bwd_rows_reg_kernel<NE, ACT_RELU>), grid, dim3(256), 0, s, dy, lddy, x, ldx, stats, gamma, beta, mask, dx, lddx, dyeff, M); break;
        default: hipLaunchKernelGGL((ln_bwd_rows_reg_kernel<NE, ACT_NONE>), grid, dim3(256), 0, s, dy, lddy, x, ldx, stats, gamma, beta, mask, dx, lddx, dyeff, M); break;
    }
}
int ln_bwd_rows(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                int act, const float* mask, float* dx, int lddx, float* dyeff, int M, int D, hipStream_t s) {
    if (act == ACT_SILU || act == ACT_RELU || act == ACT_NONE) {
        switch (D) {
            case 256: launch_ln_bwd_reg<4>(dy, lddy, x, ldx, stats, gamma, beta, act, mask, dx, lddx, dyeff, M, s); AID_LAUNCH_CHECK(); return AID_OK;
            case 512: launch_ln_bwd_reg<8>(dy, lddy, x, ldx, stats, gamma, beta, act, mask, dx, lddx, dyeff, M, s); AID_LAUNCH_CHECK(); return AID_OK;
            case 1024: launch_ln_bwd_reg<16>(dy, lddy, x, ldx, stats, gamma, beta, act, mask, dx, lddx, dyeff, M, s); AID_LAUNCH_CHECK(); return AID_OK;
            default: break;
        }
    }
    hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3((M + RPB - 1) / RPB), dim3(256), 0, s, dy, lddy, x, ldx, stats, gamma, beta, act,
                       mask, dx, lddx, dyeff, M, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void __launch_bounds__(64 * CS_WAVES) ln_param_grad_kernel(const float* __restrict__ dyeff, const float* __restrict__ x, int ldx,
                                     const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     int M, int D) {
    __shared__ float rg[CS_WAVES][64], rb[CS_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < D) for (int r = w; r < M; r += 4 * CS_WAVES) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * CS_WAVES;
            if (rr < M) {
                const float e = dyeff[(size_t)rr * D + c];
                sg[u] += e * (x[(size_t)rr * ldx + c] - stats[2 * rr]) * stats[2 * rr + 1];
                sb[u] += e;
            }
        }
    }
    rg[w][lane] = (sg[0] + sg[1]) + (sg[2] + sg[3]); rb[w][lane] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    __syncthreads();
    if (w == 0 && c < D) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < CS_WAVES; ++i) { tg += rg[i][lane]; tb += rb[i][lane]; }
        dgamma[c] += tg; dbeta[c] += tb;
    }
}
int ln_param_grad(const float* dyeff, const float* x, int ldx, const float* stats, float* dgamma, float* dbeta, int M, int D, hipStream_t s,
                  Workspace* ws) {
    if (ws && M > 2 * SLAB) {
        const int slabs = (M + SLAB - 1) / SLAB;
        const size_t mark = ws->mark();
        float* p0 = ws->f32((size_t)slabs * D); float* p1 = ws->f32((size_t)slabs * D);
        if (!ws->overflow) {
            hipLaunchKernelGGL(colsum_slab_kernel, dim3((D + 63) / 64, slabs), dim3(64 * CS_WAVES), 0, s, dyeff, D, x, ldx, stats, p0, p1, M, D);
            hipLaunchKernelGGL(colsum_fold_kernel, dim3((D + 255) / 256), dim3(256), 0, s, p0, slabs, dgamma, D, 1);
            hipLaunchKernelGGL(colsum_fold_kernel, dim3((D + 255) / 256), dim3(256), 0, s, p1, slabs, dbeta, D, 1);
            AID_LAUNCH_CHECK(); ws->release(mark); return AID_OK;
        }
        ws->overflow = false; ws->release(mark);
    }
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((D + 63) / 64), dim3(64 * CS_WAVES), 0, s, dyeff, x, ldx, stats, dgamma, dbeta, M, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// ---------------------------------------------------------------------------------------------- adaLN dual backward
__global__ void adaln_bwd_dual_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ stats,
                                      const float* __restrict__ mod, int ldmod, const float* __restrict__ mu_nx,
                                      const float* __restrict__ hd, const float* __restrict__ lam_nxd,
                                      const float* __restrict__ mu_in, const float* __restrict__ lam_in,
                                      float* __restrict__ mu_out, float* __restrict__ lam_out, float* __restrict__ dmod,
                                      int lddmod, int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * RPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    const size_t o = (size_t)row * D;
    const float* hr = h + (size_t)row * ldh; const float* sc = mod + (size_t)row * ldmod;
    const bool dual = lam_nxd != nullptr;
    float s_hd = 0.f, s_xhd = 0.f, s_mg = 0.f, s_xmg = 0.f, s_la = 0.f, s_xla = 0.f, s_lahd = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xh = (hr[c] - mean) * r, one_s = 1.f + sc[c];
        if (mu_nx) { const float mg = mu_nx[o + c] * one_s; s_mg += mg; s_xmg += mg * xh; }
        if (dual) {
            const float d = hd[o + c], la = lam_nxd[o + c] * one_s;
            s_hd += d; s_xhd += xh * d; s_la += la; s_xla += xh * la; s_lahd += la * d;
        }
    }
    const float invD = 1.f / (float)D;
    s_mg = wave_sum(s_mg) * invD; s_xmg = wave_sum(s_xmg) * invD;
    float p = 0.f, q = 0.f, A = 0.f;
    if (dual) {
        s_hd = wave_sum(s_hd) * invD; s_xhd = wave_sum(s_xhd) * invD; s_la = wave_sum(s_la) * invD;
        s_xla = wave_sum(s_xla) * invD; s_lahd = wave_sum(s_lahd) * invD;
        p = s_xhd; q = s_xla; A = s_lahd - s_la * s_hd;   // mean(lam_a * (hd - mean hd))
    }
    for (int c = lane; c < D; c += 64) {
        const float xh = (hr[c] - mean) * r, one_s = 1.f + sc[c];
        float mu = mu_in ? mu_in[o + c] : 0.f;
        float dscale = 0.f, dshift = 0.f;
        if (mu_nx) {
            const float m = mu_nx[o + c];
            mu += r * (m * one_s - s_mg - xh * s_xmg);
            dscale = m * xh; dshift = m;
        }
        if (dual) {
            const float d = hd[o + c], ln = lam_nxd[o + c], la = ln * one_s;
            const float hc = d - s_hd, lc = la - s_la;
            const float a = r * (hc - xh * p);                     // T(h)[hd]
            dscale += ln * a;
            mu += -(r * r) * (xh * (A - 3.f * p * q) + q * hc + p * lc);
            lam_out[o + c] = (lam_in ? lam_in[o + c] : 0.f) + r * (lc - xh * q);
        }
        mu_out[o + c] = mu;
        if (dmod) { dmod[(size_t)row * lddmod + c] = dscale; dmod[(size_t)row * lddmod + D + c] = dshift; }
    }
}
// The same with the row in registers (D = 256 NV: a lane holds NV float4 of each operand): every load is issued before the first use, the rows
// are read once, and the seven row statistics are folded together (fold8: 10 cross-lane moves + 7 broadcasts instead of 7 x 6 dependent
// ones).  At 256 rows this kernel is one launch-latency chain per adaLN of the backward passes (39 per ELBO step): 11.3 -> ~6 us.
template <int NV>
__global__ void __launch_bounds__(256) adaln_bwd_dual_reg_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ stats,
                                                                 const float* __restrict__ mod, int ldmod, const float* __restrict__ mu_nx,
                                                                 const float* __restrict__ hd, const float* __restrict__ lam_nxd,
                                                                 const float* __restrict__ mu_in, const float* __restrict__ lam_in,
                                                                 float* __restrict__ mu_out, float* __restrict__ lam_out, float* __restrict__ dmod,
                                                                 int lddmod, int M) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, row = blockIdx.x * RPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t o = (size_t)row * D;
    const bool dual = lam_nxd != nullptr;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    f4 hv[NV], sv[NV], mv[NV], dv[NV], lv[NV], mi[NV], li[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        hv[i] = reinterpret_cast<const f4*>(h + (size_t)row * ldh)[q];
        sv[i] = reinterpret_cast<const f4*>(mod + (size_t)row * ldmod)[q];
        mv[i] = mu_nx ? reinterpret_cast<const f4*>(mu_nx + o)[q] : z4;
        dv[i] = dual ? reinterpret_cast<const f4*>(hd + o)[q] : z4;
        lv[i] = dual ? reinterpret_cast<const f4*>(lam_nxd + o)[q] : z4;
        mi[i] = mu_in ? reinterpret_cast<const f4*>(mu_in + o)[q] : z4;
        li[i] = (dual && lam_in) ? reinterpret_cast<const f4*>(lam_in + o)[q] : z4;
    }
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // s_mg, s_xmg, s_hd, s_xhd, s_la, s_xla, s_lahd
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (hv[i][e] - mean) * r, one_s = 1.f + sv[i][e];
            const float mg = mv[i][e] * one_s; acc[0] += mg; acc[1] += mg * xh;
            const float d = dv[i][e], la = lv[i][e] * one_s;
            acc[2] += d; acc[3] += xh * d; acc[4] += la; acc[5] += xh * la; acc[6] += la * d;
        }
    const float tot = fold8(acc, lane);
    const float invD = 1.f / (float)D;
    const float s_mg = __shfl(tot, 0, 64) * invD, s_xmg = __shfl(tot, 8, 64) * invD, s_hd = __shfl(tot, 16, 64) * invD, s_xhd = __shfl(tot, 24, 64) * invD,
                s_la = __shfl(tot, 32, 64) * invD, s_xla = __shfl(tot, 40, 64) * invD, s_lahd = __shfl(tot, 48, 64) * invD;
    const float p = s_xhd, q = s_xla, A = s_lahd - s_la * s_hd;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f4 mo, lo, dsc, dsh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (hv[i][e] - mean) * r, one_s = 1.f + sv[i][e];
            float mu = mi[i][e], dscale = 0.f, dshift = 0.f;
            if (mu_nx) {
                const float m = mv[i][e];
                mu += r * (m * one_s - s_mg - xh * s_xmg);
                dscale = m * xh; dshift = m;
            }
            lo[e] = 0.f;
            if (dual) {
                const float d = dv[i][e], ln = lv[i][e], la = ln * one_s;
                const float hc = d - s_hd, lc = la - s_la;
                const float a = r * (hc - xh * p);
                dscale += ln * a;
                mu += -(r * r) * (xh * (A - 3.f * p * q) + q * hc + p * lc);
                lo[e] = li[i][e] + r * (lc - xh * q);
            }
            mo[e] = mu; dsc[e] = dscale; dsh[e] = dshift;
        }
        const int qd = lane + 64 * i;
        reinterpret_cast<f4*>(mu_out + o)[qd] = mo;
        if (dual) reinterpret_cast<f4*>(lam_out + o)[qd] = lo;
        if (dmod) { reinterpret_cast<f4*>(dmod + (size_t)row * lddmod)[qd] = dsc; reinterpret_cast<f4*>(dmod + (size_t)row * lddmod + D)[qd] = dsh; }
    }
}
int adaln_bwd_dual(const float* h, int ldh, const float* stats, const float* mod, int ldmod, const float* mu_nx,
                   const float* hd, const float* lam_nxd, const float* mu_in, const float* lam_in, float* mu_out,
                   float* lam_out, float* dmod, int lddmod, int M, int D, hipStream_t s) {
    const uintptr_t al = (uintptr_t)h | (uintptr_t)mod | (uintptr_t)mu_nx | (uintptr_t)hd | (uintptr_t)lam_nxd | (uintptr_t)mu_in | (uintptr_t)lam_in |
                         (uintptr_t)mu_out | (uintptr_t)lam_out | (uintptr_t)dmod;
    if ((D == 256 || D == 512 || D == 1024) && (al & 15) == 0 && ldh % 4 == 0 && ldmod % 4 == 0 && lddmod % 4 == 0 && (!lam_nxd || (hd && lam_out))) {
        const dim3 grid((M + RPB - 1) / RPB);
        if (D == 256) hipLaunchKernelGGL(adaln_bwd_dual_reg_kernel<1>, grid, dim3(256), 0, s, h, ldh, stats, mod, ldmod, mu_nx, hd, lam_nxd, mu_in, lam_in, mu_out, lam_out, dmod, lddmod, M);
        else if (D == 512) hipLaunchKernelGGL(adaln_bwd_dual_reg_kernel<2>, grid, dim3(256), 0, s, h, ldh, stats, mod, ldmod, mu_nx, hd, lam_nxd, mu_in, lam_in, mu_out, lam_out, dmod, lddmod, M);
        else hipLaunchKernelGGL(adaln_bwd_dual_reg_kernel<4>, grid, dim3(256), 0, s, h, ldh, stats, mod, ldmod, mu_nx, hd, lam_nxd, mu_in, lam_in, mu_out, lam_out, dmod, lddmod, M);
        AID_LAUNCH_CHECK(); return AID_OK;
    }
    hipLaunchKernelGGL(adaln_bwd_dual_kernel, dim3((M + RPB - 1) / RPB), dim3(256), 0, s, h, ldh, stats, mod, ldmod, mu_nx, hd,
                       lam_nxd, mu_in, lam_in, mu_out, lam_out, dmod, lddmod, M, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void adaln_tangent_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ stats,
                                     const float* __restrict__ mod, int ldmod, const float* __restrict__ hd,
                                     float* __restrict__ nxd, int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * RPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    const size_t o = (size_t)row * D;
    const float* hr = h + (size_t)row * ldh; const float* sc = mod + (size_t)row * ldmod;
    float s_hd = 0.f, s_xhd = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = hd[o + c]; s_hd += d; s_xhd += (hr[c] - mean) * r * d; }
    s_hd = wave_sum(s_hd) / (float)D; s_xhd = wave_sum(s_xhd) / (float)D;
    for (int c = lane; c < D; c += 64) {
        const float xh = (hr[c] - mean) * r;
        nxd[o + c] = r * (hd[o + c] - s_hd - xh * s_xhd) * (1.f + sc[c]);
    }
}
int adaln_tangent(const float* h, int ldh, const float* stats, const float* mod, int ldmod, const float* hd, float* nxd, int M,
                  int D, hipStream_t s) {
    hipLaunchKernelGGL(adaln_tangent_kernel, dim3((M + RPB - 1) / RPB), dim3(256), 0, s, h, ldh, stats, mod, ldmod, hd, nxd, M, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void act_bwd_dual_kernel(const float* __restrict__ pre, float* __restrict__ mu, float* __restrict__ lam,
                                    const float* __restrict__ xd, int act, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = pre[i], g1 = act_grad(x, act);
        float m = mu[i] * g1;
        if (lam) { const float l = lam[i]; m += l * xd[i] * act_grad2(x, act); lam[i] = l * g1; }
        mu[i] = m;
    }
}
int act_bwd_dual(const float* pre, float* mu, float* lam, const float* xd, int act, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(act_bwd_dual_kernel, dim3(blocks), dim3(256), 0, s, pre, mu, lam, xd, act, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// ---------------------------------------------------------------------------------------------- single-workgroup reductions
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < 64) { t = (threadIdx.x < (blockDim.x >> 6)) ? sh[threadIdx.x] : 0.f; t = wave_sum(t); }
    __syncthreads();
    return t;  // valid in wave 0
}

__global__ void head_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ d_score, const float* __restrict__ mult,
                                const float* __restrict__ t, int mode, const float* __restrict__ rawd, float coef,
                                float* __restrict__ mu_raw, float* __restrict__ lam_raw, float* __restrict__ dmult, int M, int L) {
    __shared__ float sh[16];
    float acc = 0.f;
    const float mlt = mult[0];
    for (int i = threadIdx.x; i < M * L; i += blockDim.x) {
        const int row = i / L;
        const float rv = raw[i];
        const float inside = (rv >= -10.f && rv <= 10.f) ? 1.f : 0.f;
        const bool cont = mode == 1 || (mode == 2 && t[row] <= 1.0f);
        const float tw = cont ? sqrtf(1.0f / (1e-5f + t[row])) : 1.f;
        if (rawd) {
            lam_raw[i] = coef * inside * mlt * tw; mu_raw[i] = 0.f;
            acc += coef * inside * tw * rawd[i];
        } else {
            const float ds = d_score ? d_score[i] : 1.f;
            mu_raw[i] = ds * mlt * tw * inside;
            acc += ds * fminf(fmaxf(rv, -10.f), 10.f) * tw;
        }
    }
    const float tot = block_sum_1024(acc, sh);
    if (dmult && threadIdx.x == 0) dmult[0] += tot;
}
int head_bwd(const float* raw, const float* d_score, const float* mult, const float* t, int mode, const float* rawd, float coef,
             float* mu_raw, float* lam_raw, float* dmult, int M, int L, hipStream_t s) {
    hipLaunchKernelGGL(head_bwd_kernel, dim3(1), dim3(1024), 0, s, raw, d_score, mult, t, mode, rawd, coef, mu_raw, lam_raw, dmult, M, L);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void sinusoid_bwd_kernel(const float* __restrict__ dsin, const float* __restrict__ t, const float* __restrict__ fs,
                                    float tmul, int mode, float* __restrict__ dfs, int M, int dim) {
    __shared__ float sh[16];
    const int half = dim / 2;
    const float negc = -(float)(9.210340371976184 / (double)(half - 1));
    float acc = 0.f;
    for (int i = threadIdx.x; i < M * half; i += blockDim.x) {
        const int row = i / half, k = i % half;
        float tv = t[row];
        const bool cont = mode == 1 || (mode == 2 && tv <= 1.0f);
        if (cont) tv *= tmul;
        const float base = tv * expf((float)k * negc);   // d(arg)/d(fs)
        const float a = base * fs[0];
        acc += (dsin[(size_t)row * dim + k] * cosf(a) - dsin[(size_t)row * dim + half + k] * sinf(a)) * base;
    }
    const float tot = block_sum_1024(acc, sh);
    if (threadIdx.x == 0) dfs[0] += tot;
}
int sinusoid_bwd(const float* dsin, const float* t, const float* fs, float tmul, int mode, float* dfs, int M, int dim, hipStream_t s) {
    hipLaunchKernelGGL(sinusoid_bwd_kernel, dim3(1), dim3(1024), 0, s, dsin, t, fs, tmul, mode, dfs, M, dim);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void dot_accum_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float scale, float* __restrict__ out) {
    __shared__ float sh[16];
    typedef float f4 __attribute__((ext_vector_type(4)));
    float acc = 0.f;
    size_t done = 0;
    if (b && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0)) {       // 16-B loads, four in flight per thread (131 072 elements were 128 dependent 4-B loads: 55 us)
        const size_t n4 = n >> 2;
        f4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (size_t i = threadIdx.x; i < n4; i += blockDim.x) s4 += reinterpret_cast<const f4*>(a)[i] * reinterpret_cast<const f4*>(b)[i];
        acc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        done = n4 << 2;
    }
    for (size_t i = done + threadIdx.x; i < n; i += blockDim.x) acc += a[i] * (b ? b[i] : 1.f);
    const float tot = block_sum_1024(acc, sh);
    if (threadIdx.x == 0) out[0] += scale * tot;
}
int dot_accum(const float* a, const float* b, size_t n, float scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dot_accum_kernel, dim3(1), dim3(1024), 0, s, a, b, n, scale, out);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void scale_by_ptr_kernel(const float* __restrict__ a, const float* __restrict__ sc, float* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float k = sc[0];
    for (; i < n; i += stride) o[i] = k * a[i];
}
int scale_by_ptr(const float* a, const float* scalar, float* out, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(scale_by_ptr_kernel, dim3(blocks), dim3(256), 0, s, a, scalar, out, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] += b[i];
}
int add_inplace(float* a, const float* b, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, s, a, b, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void gp_weights_kernel(const float* __restrict__ g, float* __restrict__ w, float* __restrict__ pen, int B, int L) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float n2 = 0.f;
        for (int l = 0; l < L; ++l) { const float v = g[(size_t)b * L + l]; n2 += v * v; }
        const float nrm = sqrtf(n2);
        acc += (nrm - 1.f) * (nrm - 1.f);
        const float k = nrm > 0.f ? (2.0f / (float)B) * (nrm - 1.f) / nrm : 0.f;
        for (int l = 0; l < L; ++l) w[(size_t)b * L + l] = k * g[(size_t)b * L + l];
    }
    const float tot = block_sum_1024(acc, sh);
    if (threadIdx.x == 0) pen[0] = tot / (float)B;
}
int gp_weights(const float* g, float* w, float* pen, int B, int L, hipStream_t s) {
    hipLaunchKernelGGL(gp_weights_kernel, dim3(1), dim3(1024), 0, s, g, w, pen, B, L);
    AID_LAUNCH_CHECK(); return AID_OK;
}

}  // namespace aid
