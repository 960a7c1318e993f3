#include "rowwise.h"
#include <algorithm>

namespace aid {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves per 256-thread workgroup, one row per wave

__device__ __forceinline__ void row_stats(const float* __restrict__ x, int D, int lane, float& mean, float& rstd) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += x[c];
    mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = x[c] - mean; v += d * d; }
    const float var = wave_sum(v) / (float)D;
    rstd = 1.0f / sqrtf(var + 1e-5f);
}

__global__ void sinusoid_kernel(const float* __restrict__ t, const float* __restrict__ fs, float tmul, int mode,
                                float* __restrict__ out, int M, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= M * half) return;
    const int row = i / half, k = i % half;
    float tv = t[row];
    const bool cont = mode == 1 || (mode == 2 && tv <= 1.0f);
    if (cont) tv = tv * tmul;
    const float negc = -(float)(9.210340371976184 / (double)(half - 1));  // -ln(1e4)/(half-1)
    const float f = expf((float)k * negc) * fs[0];
    const float a = tv * f;
    out[(size_t)row * dim + k] = sinf(a);
    out[(size_t)row * dim + half + k] = cosf(a);
}
int sinusoid_embed(const float* t, const float* fs, float tmul, int mode, float* out, int M, int dim, hipStream_t s) {
    const int n = M * (dim / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, fs, tmul, mode, out, M, dim);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void ln_act_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                              const float* __restrict__ beta, int act, const float* __restrict__ mask,
                              float* __restrict__ y, int ldy, float* __restrict__ stats, int M, int D, const float* __restrict__ add) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float mean, rstd; row_stats(xr, D, lane, mean, rstd);
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    for (int c = lane; c < D; c += 64) {
        float v = (xr[c] - mean) * rstd;
        if (gamma) v = v * gamma[c] + beta[c];
        v = act_apply(v, act);
        if (mask) v *= mask[(size_t)row * D + c];
        if (add) v += add[(size_t)row * D + c];
        y[(size_t)row * ldy + c] = v;
    }
}
template <int NV>
__global__ void ln_act_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  int act, const float* __restrict__ mask, float* __restrict__ y, int ldy, float* __restrict__ stats, int M,
                                  const float* __restrict__ add) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    const f4* xr = reinterpret_cast<const f4*>(x + (size_t)row * ldx);
    f4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xr[lane + 64 * i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    const f4* g4 = reinterpret_cast<const f4*>(gamma); const f4* b4 = reinterpret_cast<const f4*>(beta);
    const f4* m4 = mask ? reinterpret_cast<const f4*>(mask + (size_t)row * D) : nullptr;
    const f4* a4 = add ? reinterpret_cast<const f4*>(add + (size_t)row * D) : nullptr;
    f4* yr = reinterpret_cast<f4*>(y + (size_t)row * ldy);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f4 o, gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f}, mm = {1.f, 1.f, 1.f, 1.f}, aa = {0.f, 0.f, 0.f, 0.f};
        if (gamma) { gg = g4[lane + 64 * i]; bb = b4[lane + 64 * i]; }
        if (m4) mm = m4[lane + 64 * i];
        if (a4) aa = a4[lane + 64 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = act_apply((v[i][j] - mean) * rstd * gg[j] + bb[j], act) * mm[j] + aa[j];
        yr[lane + 64 * i] = o;
    }
}
int ln_act(const float* x, int ldx, const float* gamma, const float* beta, int act, const float* mask, float* y, int ldy,
           float* stats, int M, int D, hipStream_t s, const float* add) {
    const bool al = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mask | (uintptr_t)add) & 15) == 0 && ldx % 4 == 0 && ldy % 4 == 0;
    if (al && (D == 256 || D == 512 || D == 1024 || D == 2048)) {
        const dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        if (D == 256) hipLaunchKernelGGL(ln_act_vec_kernel<1>, grid, dim3(256), 0, s, x, ldx, gamma, beta, act, mask, y, ldy, stats, M, add);
        else if (D == 512) hipLaunchKernelGGL(ln_act_vec_kernel<2>, grid, dim3(256), 0, s, x, ldx, gamma, beta, act, mask, y, ldy, stats, M, add);
        else if (D == 1024) hipLaunchKernelGGL(ln_act_vec_kernel<4>, grid, dim3(256), 0, s, x, ldx, gamma, beta, act, mask, y, ldy, stats, M, add);
        else hipLaunchKernelGGL(ln_act_vec_kernel<8>, grid, dim3(256), 0, s, x, ldx, gamma, beta, act, mask, y, ldy, stats, M, add);
        AID_LAUNCH_CHECK(); return AID_OK;
    }
    hipLaunchKernelGGL(ln_act_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, x, ldx, gamma, beta,
                       act, mask, y, ldy, stats, M, D, add);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// Linear(K <= 32) -> LayerNorm -> activation in ONE pass (first layer of the decoder / reward / policy / value / dynamics nets: K is the
// latent or observation dim).  The GEMM is output-write bound at K = 17 and the LayerNorm after it re-read and re-wrote the same [M, D]
// tensor; here a wave owns 4 rows at a time: W^T [K][D] sits in LDS, the wave's x rows are broadcast from LDS, the D = 256 NV outputs of a
// row stay in registers through bias, (optional store of the pre-LN values and statistics for a backward pass), two-pass LayerNorm,
// affine, activation, mask and skip-add, and y is written once.  M x D floats of traffic instead of 3-4x that.
// (the activation is a template parameter: with a run-time switch the 64 unrolled epilogue elements inline every activation's math and the
// kernel outgrows the instruction cache — 24 000 instructions, 120 us instead of 30)
template <int NV, int ACT>
__global__ void __launch_bounds__(256) lin_smallk_ln_act_kernel(const float* __restrict__ x, int ldx, int K, const float* __restrict__ W,
                                                                const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mask,
                                                                const float* __restrict__ add, float* __restrict__ lin, float* __restrict__ y,
                                                                int ldy, float* __restrict__ stats, int M) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int D = 256 * NV, XR = 4, KP = 32;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float* Wt = lsm;                        // [K][D]
    float* xs = lsm + (size_t)K * D;        // [4 waves][XR rows][KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K * D; i += 256) { const int c = i / K, k = i - c * K; Wt[k * D + c] = W[i]; }
    __syncthreads();
    float* xw = xs + wave * XR * KP;
    const int nblk = (M + XR - 1) / XR;
    // the x values of a block are fetched one block ahead (registers), so that their latency hides behind the previous block's FMAs: with
    // 70 KB of LDS there are two waves per SIMD and nobody else to hide a load behind
    float xn[XR * KP / 64];
    auto fetch = [&](int blk) {
        const int r0 = blk * XR;
#pragma unroll
        for (int h = 0; h < XR * KP / 64; ++h) {
            const int r = h * 2 + (lane >> 5), k = lane & 31;
            const bool ok = k < K && r0 + r < M && blk < nblk;
            xn[h] = x[ok ? (size_t)(r0 + r) * ldx + k : 0];
            if (!ok) xn[h] = 0.f;
        }
    };
    fetch(blockIdx.x * 4 + wave);
    for (int blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4) {
        const int r0 = blk * XR;
        // this wave's XR rows of x -> LDS (lane = r * KP + k for the first XR*KP lanes; KP = 32 so two rows per 64 lanes)
#pragma unroll
        for (int h = 0; h < XR * KP / 64; ++h) {
            const int r = h * 2 + (lane >> 5), k = lane & 31;
            xw[r * KP + k] = xn[h];
        }
        fetch(blk + gridDim.x * 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();   // other lanes' x values are read below
        f4 acc[XR][NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f4 b = bias ? reinterpret_cast<const f4*>(bias)[lane + 64 * i] : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < XR; ++r) acc[r][i] = b;
        }
        for (int k = 0; k < K; ++k) {
            float xv[XR];
#pragma unroll
            for (int r = 0; r < XR; ++r) xv[r] = xw[r * KP + k];          // same address in every lane: LDS broadcast
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const f4 w = reinterpret_cast<const f4*>(Wt + (size_t)k * D)[lane + 64 * i];
#pragma unroll
                for (int r = 0; r < XR; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r][i][j] = fmaf(xv[r], w[j], acc[r][i][j]);
            }
        }
        __builtin_amdgcn_wave_barrier();          // all reads of xw done before the next block overwrites it
        // LayerNorm statistics of the XR rows together: the XR reduction chains are independent, so their latencies overlap
        // (one row at a time behind 12 dependent ds_bpermute steps they were ~30 % of the block's cycles at two waves per SIMD)
        float mean[XR], rstd[XR];
        {
            float sm[XR];
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                sm[r] = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) sm[r] += (acc[r][i][0] + acc[r][i][1]) + (acc[r][i][2] + acc[r][i][3]);
            }
#pragma unroll
            for (int r = 0; r < XR; ++r) sm[r] = wave_sum(sm[r]);
#pragma unroll
            for (int r = 0; r < XR; ++r) { mean[r] = sm[r] / (float)D; sm[r] = 0.f; }
#pragma unroll
            for (int r = 0; r < XR; ++r)
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = acc[r][i][j] - mean[r]; sm[r] += d * d; }
#pragma unroll
            for (int r = 0; r < XR; ++r) sm[r] = wave_sum(sm[r]);
#pragma unroll
            for (int r = 0; r < XR; ++r) rstd[r] = 1.0f / sqrtf(sm[r] / (float)D + 1e-5f);
        }
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int row = r0 + r;
            if (row >= M) break;                                         // wave-uniform
            if (lin) {
#pragma unroll
                for (int i = 0; i < NV; ++i) reinterpret_cast<f4*>(lin + (size_t)row * D)[lane + 64 * i] = acc[r][i];
            }
            if (stats && lane == 0) { stats[2 * row] = mean[r]; stats[2 * row + 1] = rstd[r]; }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f4 o, gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f}, mm = {1.f, 1.f, 1.f, 1.f}, aa = {0.f, 0.f, 0.f, 0.f};
                if (gamma) { gg = reinterpret_cast<const f4*>(gamma)[lane + 64 * i]; bb = reinterpret_cast<const f4*>(beta)[lane + 64 * i]; }
                if (mask) mm = reinterpret_cast<const f4*>(mask + (size_t)row * D)[lane + 64 * i];
                if (add) aa = reinterpret_cast<const f4*>(add + (size_t)row * D)[lane + 64 * i];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = act_apply((acc[r][i][j] - mean[r]) * rstd[r] * gg[j] + bb[j], ACT) * mm[j] + aa[j];
                reinterpret_cast<f4*>(y + (size_t)row * ldy)[lane + 64 * i] = o;
            }
        }
    }
}
// true when lin_smallk_ln_act handles the shape (callers fall back to GEMM + ln_act otherwise)
bool lin_smallk_ln_act_ok(int K, int D, int act, const void* a, const void* b, const void* c, const void* d, int ldy) {
    return (act == ACT_RELU || act == ACT_SILU) && K >= 1 && K <= 32 && (D == 256 || D == 512 || D == 1024) && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0) && ldy % 4 == 0;
}
int lin_smallk_ln_act(const float* x, int ldx, int K, const float* W, const float* bias, const float* gamma, const float* beta, int act,
                      const float* mask, const float* add, float* lin, float* y, int ldy, float* stats, int M, int D, hipStream_t s) {
    if (M <= 0) return AID_OK;
    AID_REQUIRE(act == ACT_RELU || act == ACT_SILU, AID_E_SHAPE, "lin_smallk_ln_act: activation %d not instantiated", act);
    const size_t lds = sizeof(float) * ((size_t)K * D + 4 * 4 * 32);
    const int nblk = (M + 3) / 4, grid = std::min((nblk + 3) / 4, 512);
#define LSLA(NVV, ACTV)                                                                                                        \
    do {                                                                                                                       \
        static bool done = false;                                                                                              \
        if (lds > 65536 && !done) { AID_CHECK_HIP(hipFuncSetAttribute((const void*)lin_smallk_ln_act_kernel<NVV, ACTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; } \
        hipLaunchKernelGGL((lin_smallk_ln_act_kernel<NVV, ACTV>), dim3(grid), dim3(256), lds, s, x, ldx, K, W, bias, gamma, beta, mask, add, lin, y, ldy, stats, M); \
    } while (0)
    if (act == ACT_RELU) { if (D == 256) LSLA(1, ACT_RELU); else if (D == 512) LSLA(2, ACT_RELU); else LSLA(4, ACT_RELU); }
    else                 { if (D == 256) LSLA(1, ACT_SILU); else if (D == 512) LSLA(2, ACT_SILU); else LSLA(4, ACT_SILU); }
#undef LSLA
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void ln_cond_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ temb, int ld_temb,
                               float* __restrict__ cond, float* __restrict__ sc, float* __restrict__ stats, int M, int D,
                               int x_mod, int temb_div) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)(x_mod > 0 ? row % x_mod : row) * ldx;
    if (temb_div > 0) temb += (size_t)(row / temb_div) * D - (size_t)row * ld_temb;
    float mean, rstd; row_stats(xr, D, lane, mean, rstd);
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    for (int c = lane; c < D; c += 64) {
        const float o = (xr[c] - mean) * rstd * gamma[c] + beta[c];
        const float v = temb[(size_t)row * ld_temb + c] + o;
        if (cond) cond[(size_t)row * D + c] = v;
        sc[(size_t)row * D + c] = act_apply(v, ACT_SILU);
    }
}
int ln_cond(const float* x, int ldx, const float* gamma, const float* beta, const float* temb, int ld_temb, float* cond,
            float* sc, float* stats, int M, int D, hipStream_t s, int x_mod, int temb_div) {
    hipLaunchKernelGGL(ln_cond_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, x, ldx, gamma, beta,
                       temb, ld_temb, cond, sc, stats, M, D, x_mod, temb_div);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void adaln_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mod, int ldmod,
                             float* __restrict__ y, int ldy, float* __restrict__ stats, int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    const float* mr = mod + (size_t)row * ldmod;
    float mean, rstd; row_stats(xr, D, lane, mean, rstd);
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    for (int c = lane; c < D; c += 64)
        y[(size_t)row * ldy + c] = (xr[c] - mean) * rstd * (1.0f + mr[c]) + mr[D + c];
}
// D = 256*NV: each lane keeps NV float4 of the row in registers -> one global read of x, 16-B accesses everywhere
template <int NV>
__global__ void adaln_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mod, int ldmod,
                                 float* __restrict__ y, int ldy, float* __restrict__ stats, int M) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    const f4* xr = reinterpret_cast<const f4*>(x + (size_t)row * ldx);
    const f4* sc = reinterpret_cast<const f4*>(mod + (size_t)row * ldmod);
    const f4* sh = reinterpret_cast<const f4*>(mod + (size_t)row * ldmod + D);
    f4 v[NV], a[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = xr[lane + 64 * i]; a[i] = sc[lane + 64 * i]; b[i] = sh[lane + 64 * i]; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    f4* yr = reinterpret_cast<f4*>(y + (size_t)row * ldy);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * (1.0f + a[i][j]) + b[i][j];
        yr[lane + 64 * i] = o;
    }
}
int adaln(const float* x, int ldx, const float* mod, int ldmod, float* y, int ldy, float* stats, int M, int D, hipStream_t s) {
    const bool al = (((uintptr_t)x | (uintptr_t)mod | (uintptr_t)y) & 15) == 0 && ldx % 4 == 0 && ldmod % 4 == 0 && ldy % 4 == 0;
    const dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    if (al && (D == 256 || D == 512 || D == 1024)) {
        if (D == 256) hipLaunchKernelGGL(adaln_vec_kernel<1>, grid, dim3(256), 0, s, x, ldx, mod, ldmod, y, ldy, stats, M);
        else if (D == 512) hipLaunchKernelGGL(adaln_vec_kernel<2>, grid, dim3(256), 0, s, x, ldx, mod, ldmod, y, ldy, stats, M);
        else hipLaunchKernelGGL(adaln_vec_kernel<4>, grid, dim3(256), 0, s, x, ldx, mod, ldmod, y, ldy, stats, M);
        AID_LAUNCH_CHECK(); return AID_OK;
    }
    hipLaunchKernelGGL(adaln_kernel, dim3((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, s, x, ldx, mod, ldmod, y,
                       ldy, stats, M, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void temb_combine_kernel(const float* __restrict__ ts, const float* __restrict__ c, const float* __restrict__ tscale,
                                    const float* __restrict__ t, int mode, float* __restrict__ out, int M, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * D) return;
    const int row = i / D;
    const bool cont = mode == 1 || (mode == 2 && t[row] <= 1.0f);
    out[i] = cont ? ts[i] + tscale[0] * c[i] : ts[i];
}
int temb_combine(const float* t_sin, const float* c, const float* time_scale, const float* t, int mode, float* out, int M,
                 int D, hipStream_t s) {
    const size_t n = (size_t)M * D;
    hipLaunchKernelGGL(temb_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t_sin, c, time_scale, t, mode, out, M, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void affine1_kernel(const float* __restrict__ t, float a, float b, float* __restrict__ out, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = a * t[i] + b;
}
int affine1(const float* t, float a, float b, float* out, int M, hipStream_t s) {
    hipLaunchKernelGGL(affine1_kernel, dim3((M + 255) / 256), dim3(256), 0, s, t, a, b, out, M);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void score_head_kernel(const float* __restrict__ raw, const float* __restrict__ mult, const float* __restrict__ t,
                                  int mode, float* __restrict__ score, int M, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * L) return;
    const int row = i / L;
    float v = fminf(fmaxf(raw[i], -10.f), 10.f) * mult[0];
    const bool cont = mode == 1 || (mode == 2 && t[row] <= 1.0f);
    if (cont) v *= sqrtf(1.0f / (1e-5f + t[row]));
    score[i] = v;
}
int score_head(const float* raw, const float* mult, const float* t, int mode, float* score, int M, int L, hipStream_t s) {
    hipLaunchKernelGGL(score_head_kernel, dim3((M * L + 255) / 256), dim3(256), 0, s, raw, mult, t, mode, score, M, L);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void head_p_sample_kernel(const float* __restrict__ raw, const float* __restrict__ mult, float tv, int cont,
                                     const float* __restrict__ coef, int step, const float* __restrict__ z,
                                     const float* __restrict__ eps, float* __restrict__ zo, float* __restrict__ traj, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sc = fminf(fmaxf(raw[i], -10.f), 10.f) * mult[0];
    if (cont) sc *= sqrtf(1.0f / (1e-5f + tv));
    const float zt = z[i];
    const float z0 = (zt + coef[0] * sc) * coef[1];
    float v = coef[2] * z0 + coef[3] * zt;
    if (step > 0) v += coef[4] * eps[i];
    zo[i] = v;
    if (traj) traj[i] = v;
}
// Fused last head layer + reverse step: raw = p0 . W2^T (no bias, K = D/2, N = L), then the head transform and p_sample above.
// A workgroup stages W2 ([L][K], K <= 256 per pass) and 16 rows of p0 in LDS; thread (row r = tid/16, slot j = tid%16) owns the
// outputs l = j, j+16, ... of its row.  W2 rows are padded to K+1 floats so the 16 slots of a row read 16 different banks while
// the p0 operand is a broadcast.
constexpr int H2_ROWS = 16, H2_KC = 256;
__global__ void __launch_bounds__(256) head2_p_sample_kernel(const float* __restrict__ p0, int ldp, const float* __restrict__ W2, int Kp,
                                                             const float* __restrict__ mult, float tv, int cont, const float* __restrict__ coef,
                                                             int step, const float* __restrict__ z, const float* __restrict__ eps,
                                                             float* __restrict__ zo, float* __restrict__ traj, float* __restrict__ raw_out,
                                                             int M, int L) {
    extern __shared__ float h2s[];
    float* ws = h2s;                                  // [L][H2_KC + 1]
    float* ps = h2s + (size_t)L * (H2_KC + 1);        // [H2_ROWS][H2_KC + 1]
    const int tid = threadIdx.x, r = tid >> 4, j = tid & 15;
    const int row0 = blockIdx.x * H2_ROWS, row = row0 + r;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int k0 = 0; k0 < Kp; k0 += H2_KC) {
        const int kc = min(H2_KC, Kp - k0);
        __syncthreads();
        for (int i = tid; i < L * kc; i += 256) { const int l = i / kc, k = i % kc; ws[l * (H2_KC + 1) + k] = W2[(size_t)l * Kp + k0 + k]; }
        for (int i = tid; i < H2_ROWS * kc; i += 256) { const int rr = i / kc, k = i % kc;
            ps[rr * (H2_KC + 1) + k] = (row0 + rr < M) ? p0[(size_t)(row0 + rr) * ldp + k0 + k] : 0.f; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int l = j + 16 * q;
            if (l < L) {
                const float* w = ws + l * (H2_KC + 1); const float* pr = ps + r * (H2_KC + 1);
                float a = acc[q];
                for (int k = 0; k < kc; ++k) a = fmaf(pr[k], w[k], a);
                acc[q] = a;
            }
        }
    }
    if (row >= M) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int l = j + 16 * q;
        if (l >= L) continue;
        const size_t i = (size_t)row * L + l;
        const float mine = acc[q];
        if (raw_out) raw_out[i] = mine;
        float sc = fminf(fmaxf(mine, -10.f), 10.f) * mult[0];
        if (cont) sc *= sqrtf(1.0f / (1e-5f + tv));
        const float zt = z[i];
        const float z0 = (zt + coef[0] * sc) * coef[1];
        float v = coef[2] * z0 + coef[3] * zt;
        if (step > 0) v += coef[4] * eps[i];
        zo[i] = v;
        if (traj) traj[i] = v;
    }
}
// The same with one wave per row (K = 256 KV, 16-B aligned): the lanes split K, each lane holds its float4 of the p0 row and dots it with
// 8 W2 rows at a time, fold8 produces the 8 outputs, and lane 8 c applies the head transform + reverse step to output l0 + c.  M waves
// instead of M / 16 workgroups walking K serially: 16.5 -> 5 us per reverse step at 256-768 rows.
template <int KV>
__global__ void __launch_bounds__(256) head2_p_sample_wave_kernel(const float* __restrict__ p0, int ldp, const float* __restrict__ W2, int Kp,
                                                                  const float* __restrict__ mult, float tv, int cont, const float* __restrict__ coef,
                                                                  int step, const float* __restrict__ z, const float* __restrict__ eps,
                                                                  float* __restrict__ zo, float* __restrict__ traj, float* __restrict__ raw_out,
                                                                  int M, int L, NextLp nx) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float zs[4][128];                            // the new z rows of this block's four waves, for the optional next-step tail
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 4 + wv;
    const bool live = row < M;
    if (live) {
        f4 pv[KV];
#pragma unroll
        for (int j = 0; j < KV; ++j) pv[j] = reinterpret_cast<const f4*>(p0 + (size_t)row * ldp)[lane + 64 * j];
        for (int l0 = 0; l0 < L; l0 += 8) {
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f4* w = reinterpret_cast<const f4*>(W2 + (size_t)min(l0 + q, L - 1) * Kp);
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < KV; ++j) {
                    const f4 wv4 = w[lane + 64 * j];
                    a = fmaf(pv[j][0], wv4[0], a); a = fmaf(pv[j][1], wv4[1], a); a = fmaf(pv[j][2], wv4[2], a); a = fmaf(pv[j][3], wv4[3], a);
                }
                acc[q] = a;
            }
            const float mine = fold8(acc, lane);
            const int l = l0 + (lane >> 3);
            if ((lane & 7) == 0 && l < L) {
                const size_t i = (size_t)row * L + l;
                if (raw_out) raw_out[i] = mine;
                float sc = fminf(fmaxf(mine, -10.f), 10.f) * mult[0];
                if (cont) sc *= sqrtf(1.0f / (1e-5f + tv));
                const float zt = z[i];
                const float z0 = (zt + coef[0] * sc) * coef[1];
                float v = coef[2] * z0 + coef[3] * zt;
                if (step > 0) v += coef[4] * eps[i];
                zo[i] = v;
                if (traj) traj[i] = v;
                zs[wv][l] = v;
            }
        }
    }
    if (nx.Wt == nullptr) return;                           // uniform over the grid
    __syncthreads();
    if (live) adaln_lp_row<2 * KV>(zs[wv], L, nx.Wt, nx.blp, nx.mod + (size_t)row * nx.ldmod, nx.xout + (size_t)row * nx.ldxo, nx.y + (size_t)row * nx.ldy, lane);
}
bool head2_next_ok(int Kp, int L, const NextLp& nx) {
    return (Kp == 256 || Kp == 512) && nx.D == 2 * Kp && L <= 128 && nx.Wt && nx.blp && nx.mod && nx.xout && nx.y &&
           ((((uintptr_t)nx.Wt | (uintptr_t)nx.blp | (uintptr_t)nx.mod | (uintptr_t)nx.xout | (uintptr_t)nx.y) & 15) == 0) && nx.ldmod % 4 == 0 && nx.ldxo % 4 == 0 &&
           nx.ldy % 4 == 0;
}
int head2_p_sample(const float* p0, int ldp, const float* W2, int Kp, const float* mult, float t_value, int continuous, const float* coef,
                   int step, const float* z, const float* eps, float* z_out, float* traj_out, float* raw_out, int M, int L, hipStream_t s, const NextLp* next) {
    AID_REQUIRE(L >= 1 && L <= 128, AID_E_SHAPE, "head2_p_sample: latent_dim %d outside [1, 128]", L);
    NextLp nx; memset(&nx, 0, sizeof(nx));
    if (next) nx = *next;
    if ((Kp == 256 || Kp == 512) && ldp % 4 == 0 && ((((uintptr_t)p0 | (uintptr_t)W2) & 15) == 0)) {
        AID_REQUIRE(!next || head2_next_ok(Kp, L, nx), AID_E_SHAPE, "head2_p_sample: next-step tail unsupported (Kp=%d D=%d)", Kp, nx.D);
        if (Kp == 256) hipLaunchKernelGGL(head2_p_sample_wave_kernel<1>, dim3((M + 3) / 4), dim3(256), 0, s, p0, ldp, W2, Kp, mult, t_value, continuous, coef,
                                          step, z, eps, z_out, traj_out, raw_out, M, L, nx);
        else hipLaunchKernelGGL(head2_p_sample_wave_kernel<2>, dim3((M + 3) / 4), dim3(256), 0, s, p0, ldp, W2, Kp, mult, t_value, continuous, coef,
                                step, z, eps, z_out, traj_out, raw_out, M, L, nx);
        AID_LAUNCH_CHECK(); return AID_OK;
    }
    AID_REQUIRE(!next, AID_E_SHAPE, "head2_p_sample: the next-step tail needs the wave kernel (Kp=%d)", Kp);
    const size_t lds = sizeof(float) * ((size_t)L * (H2_KC + 1) + (size_t)H2_ROWS * (H2_KC + 1));
    static bool attr_done = false;
    if (lds > 65536 && !attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)head2_p_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(head2_p_sample_kernel, dim3((M + H2_ROWS - 1) / H2_ROWS), dim3(256), lds, s, p0, ldp, W2, Kp, mult, t_value, continuous,
                       coef, step, z, eps, z_out, traj_out, raw_out, M, L);
    AID_LAUNCH_CHECK(); return AID_OK;
}

int head_p_sample(const float* raw, const float* mult, float t_value, int continuous, const float* coef, int step,
                  const float* z, const float* eps, float* z_out, float* traj_out, int M, int L, hipStream_t s) {
    const int n = M * L;
    hipLaunchKernelGGL(head_p_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, s, raw, mult, t_value, continuous, coef, step,
                       z, eps, z_out, traj_out, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
int fill_const(float* p, float v, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, v, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}
__global__ void mul_kernel(const float* a, const float* b, float* o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) o[i] = a[i] * b[i];
}
int mul_elem(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, s, a, b, out, n);
    AID_LAUNCH_CHECK(); return AID_OK;
}
__global__ void mul_rows_mod_kernel(const float* __restrict__ a, int a_rows, const float* __restrict__ m, float* __restrict__ o, size_t n, int D) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { const size_t r = i / D; o[i] = a[(r % a_rows) * D + i % D] * m[i]; }
}
int mul_rows_mod(const float* a, int a_rows, const float* mask, float* out, size_t R, int D, hipStream_t s) {
    const size_t n = R * D;
    if (n == 0) return AID_OK;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(mul_rows_mod_kernel, dim3(blocks), dim3(256), 0, s, a, a_rows, mask, out, n, D);
    AID_LAUNCH_CHECK(); return AID_OK;
}
int copy_f32(const float* a, float* out, size_t n, hipStream_t s) {
    if (n == 0) return AID_OK;
    AID_CHECK_HIP(hipMemcpyAsync(out, a, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return AID_OK;
}

}  // namespace aid
