// Thin-output products C[M, N <= 32] = act(alpha A[M,K] W + bias) (+ residual, + C) at 33..8192 rows, 64 <= K <= 1024: the heads of the
// small networks over the rollout's K*B = 2 560 rows (N = 6 / 2 / 1), the data gradients into their 17- / 23-wide inputs, the decoder's
// last layer at 256 rows.  As 32 x 32 MFMA tiles these are <= 80 workgroups running a 512-deep contraction each behind unaligned
// (ld = 17) operand loads: 13 - 27 us per launch, ~45 launches per rollout + BPTT.  Here W (<= 64 KB) is transposed into LDS once per
// workgroup and a wave owns two rows at a time: a lane holds its rows' A values for k = 4 lane + 256 j (16-B loads), forms its share of
// eight outputs per row and pass, and fold8 turns the 8 x 64 partial sums into 8 totals with 10 cross-lane moves.  No MFMA: the work
// is 2 M N K <= 90 MFLOP and the launch is latency, not arithmetic.  Fixed summation order (lane-strided k, then the fold): reproducible.
#include "aid_common.h"

namespace aid {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int THIN_R = 2;      // rows per wave and pass: 2 560 rows are 320 workgroups

template <int KJ, bool BKM>
__global__ void __launch_bounds__(256) gemm_thin_n_kernel(GemmArgs g) {
    constexpr int R = THIN_R;
    extern __shared__ __attribute__((aligned(16))) float Wt[];          // [N][K], k contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = g.K, N = g.N;
    int kk[KJ]; bool kin[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) { const int k = 4 * lane + 256 * j; kin[j] = k < K; kk[j] = kin[j] ? k : 0; }
    const int nblk = (g.M + R - 1) / R;
    // the first block's A values are requested before W is staged: the two latencies overlap (most launches are one block per wave)
    f4 a[R][KJ];
    auto fetch = [&](int blk) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const bool ok = kin[j] && blk < nblk && blk * R + r < g.M;
                a[r][j] = *reinterpret_cast<const f4*>(g.A + (size_t)(ok ? blk * R + r : 0) * g.lda + kk[j]);
                if (!ok) a[r][j] = (f4){0.f, 0.f, 0.f, 0.f};
            }
    };
    fetch(blockIdx.x * 4 + wave);
    const bool al = (((uintptr_t)g.B) & 15) == 0;
    const int nq = (N * K) >> 2;                                        // K % 4 == 0
    // (dense W: every float4 of the <= 64 KB block is requested before the first one is used - up to 16 per thread, one latency)
    if (!BKM && al && g.ldb == K) {                                     // W [N][K] dense: the LDS image is a flat copy
        f4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int q = tid + 256 * u; v[u] = reinterpret_cast<const f4*>(g.B)[q < nq ? q : 0]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int q = tid + 256 * u; if (q < nq) reinterpret_cast<f4*>(Wt)[q] = v[u]; }
    } else if (!BKM) {
        for (int n = wave; n < N; n += 4)
            for (int k = lane; k < K; k += 64) Wt[n * K + k] = g.B[(size_t)n * g.ldb + k];
    } else if (al && g.ldb == N) {
        // W as the layer stores it, [K][N] dense (ld = 17: rows are not 16-B aligned, the block is): flat float4 loads, one
        // division per load to find (k, n) of its first element
        f4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int q = tid + 256 * u; v[u] = reinterpret_cast<const f4*>(g.B)[q < nq ? q : 0]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = tid + 256 * u;
            if (q < nq) {
                int k = (4 * q) / N, n = 4 * q - k * N;
#pragma unroll
                for (int e = 0; e < 4; ++e) { Wt[n * K + k] = v[u][e]; if (++n == N) { n = 0; ++k; } }
            }
        }
    } else {
        for (int k = tid; k < K; k += 256) {
            const float* wr = g.B + (size_t)k * g.ldb;
#pragma unroll 8
            for (int n = 0; n < N; ++n) Wt[n * K + k] = wr[n];
        }
    }
    float bv[4];                                                // this lane's bias values, one per pass (requested before they are needed)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) { const int n = 8 * ci + (lane >> 3); bv[ci] = (g.bias && n < N) ? g.bias[n] : 0.f; }
    __syncthreads();
    for (int blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4) {
        const int r0 = blk * R;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {                        // N <= 32: at most four passes of eight columns, unrolled so that their loads overlap
            const int n0 = 8 * ci;
            if (n0 >= N) break;
            float acc[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int n = n0 + c;
                if (n >= N) continue;                            // wave-uniform: the last pass of N = 17 forms one column, not eight
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const f4 w = *reinterpret_cast<const f4*>(Wt + n * K + kk[j]);
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r][c] = fmaf(a[r][j][e], w[e], acc[r][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = fold8(acc[r], lane);                // lanes 8 c .. 8 c + 7 hold column n0 + c
                const int n = n0 + (lane >> 3), row = r0 + r;
                if ((lane & 7) == 0 && n < N && row < g.M) {
                    float v = g.alpha * t + bv[ci];
                    const size_t o = (size_t)row * g.ldc + n;
                    if (g.preact) g.preact[o] = v;
                    v = g.act == ACT_RELU ? fmaxf(v, 0.f) : (g.act == ACT_SILU ? v * __builtin_amdgcn_rcpf(1.f + exp_fast(-v)) : v);      // (the full switch, inlined per pass and row, is 4x the code)
                    if (g.residual) v += g.residual[(size_t)row * g.ldr + n];
                    if (g.accumulate) v += g.C[o];
                    g.C[o] = v;
                }
            }
        }
        fetch(blk + gridDim.x * 4);
    }
}

bool gemm_thin_n_ok(const GemmArgs& g) {
    return !g.a_kmajor && g.N >= 1 && g.N <= 32 && g.K >= 64 && g.K <= 1024 && g.K % 4 == 0 && g.lda % 4 == 0 && (((uintptr_t)g.A) & 15) == 0 &&
           g.M > 32 && g.M <= 8192 && (g.act == ACT_NONE || g.act == ACT_RELU || g.act == ACT_SILU) && !g.mask && !g.dact_pre && !g.slab && g.ksplit <= 1 && (size_t)g.N * g.K * sizeof(float) <= 65536;
}

template <int KJ>
static int launch_thin_kj(const GemmArgs& g, hipStream_t s) {
    const int nblk = (g.M + THIN_R - 1) / THIN_R, grid = (nblk + 3) / 4;
    const size_t lds = sizeof(float) * (size_t)g.N * g.K;
    if (g.b_kmajor) hipLaunchKernelGGL((gemm_thin_n_kernel<KJ, true>), dim3(grid), dim3(256), lds, s, g);
    else hipLaunchKernelGGL((gemm_thin_n_kernel<KJ, false>), dim3(grid), dim3(256), lds, s, g);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
int gemm_thin_n(const GemmArgs& g, hipStream_t s) {
    AID_REQUIRE(gemm_thin_n_ok(g), AID_E_SHAPE, "gemm_thin_n: unsupported M=%d N=%d K=%d", g.M, g.N, g.K);
    const int kj = (g.K + 255) / 256;
    if (kj <= 1) return launch_thin_kj<1>(g, s);
    if (kj <= 2) return launch_thin_kj<2>(g, s);
    return launch_thin_kj<4>(g, s);
}

}  // namespace aid
