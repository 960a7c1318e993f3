// Linear -> LayerNorm -> activation (-> dropout mask, skip add) as ONE kernel for tall inputs:  Y = act(LN(A W^T + b) * gamma + beta) * mask + add.
//
// Replaces, for the 25 600-row (and, with the perturbed copies batched, 128 000-row) activations of the gradient-free MINE estimator
// (reference: core/active_inference.py:237-242 `observation_decoder[1..2]` = Linear, LayerNorm, SiLU, Dropout inside
// FunctionSpaceEpistemicEstimator.compute_jacobian_features :940-993), the sequence  64x64-tile GEMM -> write [M,N] -> LayerNorm kernel
// reads [M,N] (+ skip) -> writes [M,N]:  here a workgroup owns 64 COMPLETE output rows (BN = the layer width, 512 or 1024), so the
// row statistics are formed from the accumulators and the normalised, activated row is the only thing written.
//
// MI355X mapping
//   * 512 threads = 8 waves; wave w owns all 64 rows x columns [w BN/8, (w+1) BN/8): 2 x (BN/256) accumulators of 32x32
//     (v_mfma_f32_32x32x2_f32; 128 accumulator registers at BN = 1024), two waves per SIMD, one workgroup per CU.
//   * K is walked in stages of BKT (16 at BN = 1024, 32 at BN = 512): one stage of the weight operand is BN x BKT x 4 B = 64 KB, double
//     buffered in LDS (128 KB) next to the 64 x BKT activation stage; per stage a wave issues 64 MFMAs (4 096 cycles of its SIMD's
//     matrix pipe) between two workgroup barriers - four times the work per barrier of the 64x64-tile kernel, whose per-K-tile barrier
//     convoy was measured to idle the pipe 28 % of the time (profiles/r01_gemm_big_stall_breakdown.txt).
//   * the weight stage is read from L2 once per 64 rows: 68 KB per 8 192 SIMD-cycles = 8.3 B/clk/CU (the 64x64 tile needs 16).
//   * LDS images are unpadded [row][BKT] with the 16-B chunk index XOR-swizzled by the row (chunk ^= (row / rows-per-256B) % chunks),
//     which makes both the b128 staging writes and the b128 fragment reads bank-conflict free (MI355X_MICROARCH.md, LDS table).
//   * register staging, prefetch distance 1: stage t+1's global loads are issued before stage t is contracted and written to the
//     other LDS buffer after it; loads are branch-free (clamped addresses, zeroed at the LDS write when beyond K).
//   * epilogue: bias; two-pass LayerNorm (mean, then centred second moment - torch's numerics) with a butterfly that folds a lane's 32
//     row partials to one row per lane in 31 cross-lane moves, cross-wave sums through LDS; affine, activation, skip; one store.
#include "aid_common.h"
#include "rowwise.h"
#include "rowln.h"

namespace aid {

typedef __attribute__((address_space(3))) void* rl_lds_vptr;
typedef const __attribute__((address_space(1))) void* rl_gbl_vptr;

template <int BN, int BKT, int ACT, bool KTAIL, int DBG = 0, bool LN = true, bool DMA = false>
__global__ void __launch_bounds__(512) gemm_rowln_kernel(RowLnArgs g) {
    constexpr int NT = 512, BM = 64, TN = BN / 256, CPR = BKT / 4;
    constexpr int A_FL = BM * BKT, B_FL = BN * BKT;
    constexpr int F4A = BM * CPR, F4B = BN * CPR, PB = F4B / NT;
    static_assert(F4B % NT == 0 && F4A <= NT && NT % F4A == 0, "stage shapes");
    using SW = RowSw<BKT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BM][BKT]
    float* Bs = smem + 2 * A_FL;        // [2][BN][BKT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wn = wave * (BN / 8);
    const int m0 = blockIdx.x * BM;
    if constexpr (!LN) {    // plain Linear (+ act, mask, add): the output may be several BN-wide column blocks (blockIdx.y); rows still whole per block
        const int n0 = blockIdx.y * BN;
        g.W += (size_t)n0 * g.ldw; g.Y += n0;
        if (g.bias) g.bias += n0;
        if (g.add) g.add += n0;
        if (g.mask) g.mask += n0;
    }
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // ---- staging maps (fixed per thread)
    const int fa = tid % F4A, a_row = fa / CPR, a_c = fa % CPR;
    const int a_gr = min(m0 + a_row, g.M - 1);                  // rows beyond M repeat the last row (computed, never stored)
    const float* a_src = g.A + (size_t)a_gr * g.lda + 4 * a_c;
    const int a_dst = SW::off(a_row, a_c);
    const bool a_owner = tid < F4A;
    const float* b_src[PB]; int b_dst[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int f = tid + NT * i, row = f / CPR, c = f % CPR;
        b_src[i] = g.W + (size_t)row * g.ldw + 4 * c; b_dst[i] = SW::off(row, c);
        if constexpr (DBG == 1) b_src[i] = g.W + (size_t)f * 4;        // timing experiment: fully coalesced (wrong) weight addresses
    }
    const int nk = (g.K + BKT - 1) / BKT;
    // DMA variant (K % BKT == 0): every staging piece is ONE global_load_lds_dwordx4 - 64 lanes x 16 B land in 1 KB of consecutive LDS, i.e. in
    // 64 / CPR whole rows of the unpadded image; a lane fetches the logical chunk that belongs at ITS LDS position (chunk ^ swizzle(row): the
    // same involution the fragment reads apply), so no VGPR staging, no ds_write and no vmcnt wait sits in the MFMA stream: a wave waits for
    // its own DMAs once, in front of the barrier that ends the stage.
    static_assert(!DMA || !KTAIL, "the DMA variant cannot zero a K tail");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int RPI = 64 / CPR;                                   // rows per DMA instruction
    const float* dma_src_b[PB]; const float* dma_src_a = nullptr;
    if constexpr (DMA) {
        const int lr = lane / CPR, lc = lane % CPR;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int row = (64 * wave + NT * p) / CPR + lr;
            dma_src_b[p] = g.W + (size_t)row * g.ldw + 4 * (lc ^ ((row / SW::RPB) % CPR));
        }
        const int rowa = (64 * wave) / CPR + lr;                    // waves below F4A / 64 own the activation pieces
        dma_src_a = g.A + (size_t)min(m0 + (rowa < BM ? rowa : 0), g.M - 1) * g.lda + 4 * (lc ^ ((rowa / SW::RPB) % CPR));
    }
    auto dma_piece = [&](int p, int kt, int buf) {
        const int k0 = kt * BKT;
        if (p == PB) {
            if (wave_u < F4A / 64) __builtin_amdgcn_global_load_lds((rl_gbl_vptr)(dma_src_a + k0), (rl_lds_vptr)(As + buf * A_FL + wave_u * RPI * BKT), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((rl_gbl_vptr)(dma_src_b[p] + k0), (rl_lds_vptr)(Bs + buf * B_FL + ((64 * wave_u + NT * p) / CPR) * BKT), 16, 0, 0);
        }
    };
    f32x4 ra, rb[PB];
    // piece p of stage kt: global -> registers (p = 0 .. PB-1: weight pieces, p = PB: the activation piece).  Branch-free: clamped k,
    // zeroed at the LDS write.
    auto load_piece = [&](int p, int kt) {
        const int k0 = kt * BKT;
        if (p == PB) {
            const int ka = (!KTAIL || k0 + 4 * a_c < g.K) ? k0 : 0;
            ra = *reinterpret_cast<const f32x4*>(a_src + ka);
        } else {
            if constexpr (DBG >= 2) { if (kt > 0) return; }             // timing experiments: no weight loads after the first stage
            const int c = (tid + NT * p) % CPR;
            const int kb = (!KTAIL || k0 + 4 * c < g.K) ? k0 : 0;
            rb[p] = *reinterpret_cast<const f32x4*>(b_src[p] + (DBG == 1 ? kb * (BN / 4) : kb));
        }
    };
    // piece p of the staged tile -> LDS buffer `buf` (p = 0 .. PB-1: weight pieces, p = PB: the activation piece)
    auto store_piece = [&](int p, int kt, int buf) {
        const int k0 = kt * BKT;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (p == PB) {
            if (a_owner) *reinterpret_cast<f32x4*>(As + buf * A_FL + a_dst) = (!KTAIL || k0 + 4 * a_c < g.K) ? ra : z;
        } else {
            const int c = (tid + NT * p) % CPR;
            *reinterpret_cast<f32x4*>(Bs + buf * B_FL + b_dst[p]) = (!KTAIL || k0 + 4 * c < g.K) ? rb[p] : z;
        }
    };
    // One stage: NM = 64 MFMAs per wave.  The hand-over work of the NEXT stage (wait for its global loads, write its PB + 1 pieces to the
    // other LDS buffer) is spread over the second half of this stage's MFMAs: an fp32 MFMA occupies the matrix pipe for 64 cycles but the
    // issue port for 8, so the same wave's ds_write / s_waitcnt issue in its shadow instead of in a phase of their own in which (both
    // waves of a SIMD being in lock-step behind the workgroup barrier) the pipe would idle.  The global loads of the next stage are
    // spread over the FIRST half the same way: issued as one burst behind the barrier, the 72 wave-instructions of a workgroup queue at
    // the CU's one address unit and every wave sits in that queue before it can issue its first MFMA (measured: 0.4 - 1.2 us per stage).
    constexpr int NG = BKT / 8, NM = NG * 4 * 2 * TN, NS = PB + 1, S0 = NM / 2, STEP = (NM - S0) / NS, LSTEP = S0 / NS;
    static_assert(STEP >= 1 && LSTEP >= 1, "more hand-over pieces than MFMA slots");
    auto stage = [&](int cur, int kt_next) {
        const float* a_s = As + cur * A_FL;
        const float* b_s = Bs + cur * B_FL;
        f32x4 av[2][2], bv[2][TN];             // [parity of g8]: fragments of group g8 + 1 are read while group g8 is contracted
#pragma unroll
        for (int i = 0; i < 2; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(a_s + SW::off(32 * i + li, h));
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(b_s + SW::off(wn + 32 * j + li, h));
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) {
            const int pc = g8 & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int idx = ((g8 * 4 + q) * 2 + i) * TN + j;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pc][i][q], bv[pc][j][q], acc[i][j], 0, 0, 0);
                        if (q == 1 && i == 0 && j == 0 && g8 + 1 < NG) {     // next group's fragments, early in this group
#pragma unroll
                            for (int i2 = 0; i2 < 2; ++i2) av[pc ^ 1][i2] = *reinterpret_cast<const f32x4*>(a_s + SW::off(32 * i2 + li, 2 * (g8 + 1) + h));
#pragma unroll
                            for (int j2 = 0; j2 < TN; ++j2) bv[pc ^ 1][j2] = *reinterpret_cast<const f32x4*>(b_s + SW::off(wn + 32 * j2 + li, 2 * (g8 + 1) + h));
                        }
                        if (idx < S0 && idx % LSTEP == 0 && idx / LSTEP < NS) {
                            if constexpr (DMA) dma_piece(idx / LSTEP, kt_next, cur ^ 1); else load_piece(idx / LSTEP, kt_next);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (!DMA && idx >= S0 && (idx - S0) % STEP == 0 && (idx - S0) / STEP < NS) {
                            if constexpr (DBG < 3) store_piece((idx - S0) / STEP, kt_next, cur ^ 1);      // DBG 3, 4: no LDS hand-over writes
                            __builtin_amdgcn_sched_barrier(0);           // pin the piece here: the scheduler otherwise sinks every write to the end
                        }
                    }
        }
    };
    if constexpr (DMA) {
#pragma unroll
        for (int p = 0; p < NS; ++p) dma_piece(p, 0, 0);
    } else {
#pragma unroll
        for (int p = 0; p < NS; ++p) load_piece(p, 0);
#pragma unroll
        for (int p = 0; p < NS; ++p) store_piece(p, 0, 0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // the last iteration re-loads its own stage and writes it to the idle buffer: nobody reads it, and the loop stays branch-free
        const int ktn = kt + 1 < nk ? kt + 1 : kt;
        stage(cur, ktn);
        if constexpr (DBG != 3) __syncthreads();     // DBG 3: no barrier either (fragment reads + MFMAs only)
    }
    // ---- epilogue.  LDS scratch (the stage buffers are dead after the loop's last barrier)
    float* part = smem;                 // [8 waves][64 rows]
    float* rowv = smem + 8 * BM;        // [64]: mean, then 1/std
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bcol[j] = g.bias ? g.bias[wn + 32 * j + li] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bcol[j];
    if constexpr (LN) {
    // accumulator register r of row-tile i is row 32 i + (r & 3) + 8 (r >> 2) + 4 h; fold index 16 i + r
    const int my_row = 32 * (li >> 4) + ((li & 15) & 3) + 8 * ((li & 15) >> 2) + 4 * h;     // the row lane li ends up owning after fold32
    float x[32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) s += acc[i][j][r];
            x[16 * i + r] = s;
        }
    part[wave * BM + my_row] = fold32(x, li);
    __syncthreads();
    if (tid < BM) { float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += part[w * BM + tid];
        rowv[tid] = s / (float)BN; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(rowv + 32 * i + 8 * rq + 4 * h);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * rq + e] -= m4[e];
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) s += acc[i][j][r] * acc[i][j][r];
            x[16 * i + r] = s;
        }
    __syncthreads();                                     // every wave has read the means before `rowv` / `part` are reused
    part[wave * BM + my_row] = fold32(x, li);
    __syncthreads();
    if (tid < BM) { float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += part[w * BM + tid];
        rowv[tid] = 1.0f / sqrtf(s / (float)BN + g.eps); }
    __syncthreads();
    }   // LN
    float gcol[TN], becol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { gcol[j] = (LN && g.gamma) ? g.gamma[wn + 32 * j + li] : 1.f; becol[j] = (LN && g.beta) ? g.beta[wn + 32 * j + li] : 0.f; }
    // 8 groups of 4 rows x TN columns per lane; the skip values of group gq + 1 are loaded before group gq is finished and stored, so
    // their latency hides behind the previous group's math and stores (one workgroup per CU: nothing else would cover it)
    float sk[2][4][TN], mk[2][4][TN];
    // branch-free skip / mask operands: an absent one is read from a valid address (element 0 of the output's first row... of `Y`'s allocation: g.Y) and dropped by a
    // select, so the loads are unconditional global loads the compiler can count (`vmcnt(N)`): behind `if (g.add)` branches it waited `vmcnt(0)` at every join and the
    // next group's prefetch was drained with the current one
    typedef const __attribute__((address_space(1))) float* gfp;
    const int add_on = g.add ? 1 : 0, mask_on = g.mask ? 1 : 0;
    const gfp addp = (gfp)(g.add ? g.add : g.Y), maskp = (gfp)(g.mask ? g.mask : g.Y);
    const size_t add_ld = g.add ? (size_t)g.ldadd : 0, mask_ld = g.mask ? (size_t)g.ldmask : 0;
    auto load_aux = [&](int gq, int pb) {
        const int i = gq >> 2, rq = gq & 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = min(m0 + 32 * i + 8 * rq + 4 * h + e, g.M - 1);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                sk[pb][e][j] = addp[(size_t)row * add_ld + (wn + 32 * j + li) * add_on];
                mk[pb][e][j] = maskp[(size_t)row * mask_ld + (wn + 32 * j + li) * mask_on];
            }
        }
    };
    load_aux(0, 0);
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
        const int i = gq >> 2, rq = gq & 3, pb = gq & 1;
        if (gq + 1 < 8) load_aux(gq + 1, pb ^ 1);
        f32x4 r4 = {1.f, 1.f, 1.f, 1.f};
        if constexpr (LN) r4 = *reinterpret_cast<const f32x4*>(rowv + 32 * i + 8 * rq + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + 32 * i + 8 * rq + 4 * h + e;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn + 32 * j + li;
                float o = LN ? rowln_act<ACT>(acc[i][j][4 * rq + e] * r4[e] * gcol[j] + becol[j]) : rowln_act<ACT>(acc[i][j][4 * rq + e]);
                o = mask_on ? o * mk[pb][e][j] : o;
                o = add_on ? o + sk[pb][e][j] : o;
                if (row < g.M) g.Y[(size_t)row * g.ldy + col] = o;
            }
        }
    }
}

static int g_rowln_dma = 1;     // 1: LDS-DMA staging where K % BKT == 0 (aid_rowln_debug(16) switches it off for A/B timing)
template <int BN, int BKT, int ACT, bool LNF>
static int launch_rowln_dma(const RowLnArgs& g, dim3 grid, hipStream_t s) {
    constexpr size_t lds = sizeof(float) * 2 * (64 * BKT + (size_t)BN * BKT);
    static bool attr_done = false;
    if (!attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rowln_kernel<BN, BKT, ACT, false, 0, LNF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_rowln_kernel<BN, BKT, ACT, false, 0, LNF, true>), grid, dim3(512), lds, s, g);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
template <int BN, int BKT, int ACT, bool KTAIL>
static int launch_rowln_t(const RowLnArgs& g, hipStream_t s) {
    // LDS-DMA staging pays at 16-deep stages (BN = 1024: 105.4 vs 102.6 TFLOP/s at 128 000 rows) and loses at 32-deep ones (BN = 512: 110.2 vs
    // 116.5), measured in one process (tools/perf/rowln_bench.py, ROWLN_AB=1): the 512-wide kernels keep the register path
    if constexpr (!KTAIL && BN == 1024) { if (g_rowln_dma) return launch_rowln_dma<BN, BKT, ACT, true>(g, dim3((g.M + 63) / 64), s); }
    constexpr size_t lds = sizeof(float) * 2 * (64 * BKT + (size_t)BN * BKT);
    static bool attr_done = false;
    if (!attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rowln_kernel<BN, BKT, ACT, KTAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_rowln_kernel<BN, BKT, ACT, KTAIL>), dim3((g.M + 63) / 64), dim3(512), lds, s, g);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
static int g_rowln_dbg = 0;     // timing experiments only (tools/perf/rowln_bench.py): 1 coalesced-but-wrong weight addresses, 2 no weight loads,
                                // 4 no weight loads and no LDS hand-over writes, 3 no barrier either.  Results are wrong by construction.
template <int BN, int BKT, int ACT, int DBG>
static int launch_rowln_dbg(const RowLnArgs& g, hipStream_t s) {
    constexpr size_t lds = sizeof(float) * 2 * (64 * BKT + (size_t)BN * BKT);
    (void)hipFuncSetAttribute((const void*)gemm_rowln_kernel<BN, BKT, ACT, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm_rowln_kernel<BN, BKT, ACT, false, DBG>), dim3((g.M + 63) / 64), dim3(512), lds, s, g);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
// 512-wide kernels with 16-deep stages use 74 KB of LDS instead of 147 (and 140 VGPRs instead of 182), so a workgroup of the reverse loop's kernel
// (gemm_mid: 80 KB, 4 waves) launched on another stream can share the CU instead of waiting for this one to retire.  Alone they are 0.2 % slower, with the
// reverse loop running underneath the step is 0.7 ms shorter (tools/perf/README.md).  Chosen per STREAM: aid_stream_hint bit 1, which the agent sets on its
// main stream while the side stream is in use.  aid_rowln_debug(34 / 35) forces it on / off process-wide for A/B timing.
static int g_rowln_bkt16 = -1;
static int g_rowln_bkt8 = 0;       // experiment (aid_rowln_debug(37 / 38)): under the stream hint the 512-wide kernels take 8-deep stages (36 KB of LDS: TWO chain workgroups fit beside one)
static int g_rowln_split1024 = 0;
static inline bool rowln_bkt16(hipStream_t s) { return g_rowln_bkt16 >= 0 ? g_rowln_bkt16 != 0 : (stream_hint_flags(s) & 2) != 0; }
template <int BN, int BKT, int ACT>
static int launch_rowln(const RowLnArgs& g, hipStream_t s) {
    if constexpr (BN == 512 && BKT == 32) {
        if (rowln_bkt16(s) && g_rowln_bkt8 && g.K % 8 == 0) return launch_rowln_t<512, 8, ACT, false>(g, s);
        if (rowln_bkt16(s) && g.K % 16 == 0) return launch_rowln_t<512, 16, ACT, false>(g, s);
    }
    if constexpr (ACT == ACT_SILU) {
        switch (g_rowln_dbg) {
            case 1: return launch_rowln_dbg<BN, BKT, ACT, 1>(g, s);
            case 2: return launch_rowln_dbg<BN, BKT, ACT, 2>(g, s);
            case 3: return launch_rowln_dbg<BN, BKT, ACT, 3>(g, s);
            case 4: return launch_rowln_dbg<BN, BKT, ACT, 4>(g, s);
            default: break;
        }
    }
    return g.K % BKT == 0 ? launch_rowln_t<BN, BKT, ACT, false>(g, s) : launch_rowln_t<BN, BKT, ACT, true>(g, s);
}

template <int ACT>
static int launch_rows_plain(const RowLnArgs& g, int N, hipStream_t s) {
    constexpr int BN = 512, BKT = 32;
    constexpr size_t lds = sizeof(float) * 2 * (64 * BKT + (size_t)BN * BKT);
    static bool attr_done = false;
    if (!attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rowln_kernel<BN, BKT, ACT, false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rowln_kernel<BN, BKT, ACT, true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const dim3 grid((g.M + 63) / 64, N / BN);
    if (rowln_bkt16(s) && g_rowln_bkt8 && g.K % 8 == 0) {
        constexpr size_t lds8 = sizeof(float) * 2 * (64 * 8 + (size_t)BN * 8);
        hipLaunchKernelGGL((gemm_rowln_kernel<BN, 8, ACT, false, 0, false>), grid, dim3(512), lds8, s, g);
        AID_LAUNCH_CHECK();
        return AID_OK;
    }
    if (rowln_bkt16(s) && g.K % 16 == 0) {
        constexpr size_t lds16 = sizeof(float) * 2 * (64 * 16 + (size_t)BN * 16);
        static bool a16 = false;
        if (!a16) { AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rowln_kernel<BN, 16, ACT, false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16)); a16 = true; }
        hipLaunchKernelGGL((gemm_rowln_kernel<BN, 16, ACT, false, 0, false>), grid, dim3(512), lds16, s, g);
        AID_LAUNCH_CHECK();
        return AID_OK;
    }
    if (g.K % BKT == 0) hipLaunchKernelGGL((gemm_rowln_kernel<BN, BKT, ACT, false, 0, false>), grid, dim3(512), lds, s, g);
    else hipLaunchKernelGGL((gemm_rowln_kernel<BN, BKT, ACT, true, 0, false>), grid, dim3(512), lds, s, g);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
// y[M,N] = act(x W^T + b) * mask + add through the same 64-row x 512-column full-K workgroups (no LayerNorm): the tall plain Linear layers
// (the MINE statistics network, the hoisted adaLN-modulation GEMM of the reverse loop).  N % 512 == 0.
bool linear_rows_ok(int M, int N, int K, int act, const float* x, int ldx, const float* W, int ldw) {
    const bool al = ((((uintptr_t)x | (uintptr_t)W) & 15) == 0) && ldx % 4 == 0 && ldw % 4 == 0 && K % 4 == 0;
    return al && M >= 4096 && N >= 512 && N % 512 == 0 && K >= 64 && (act == ACT_NONE || act == ACT_RELU || act == ACT_SILU || act == ACT_GELU);
}
int linear_rows(const float* x, int ldx, const float* W, int ldw, const float* b, int act, const float* mask, int ldmask, const float* add, int ldadd,
                float* y, int ldy, int M, int N, int K, hipStream_t s) {
    AID_REQUIRE(linear_rows_ok(M, N, K, act, x, ldx, W, ldw), AID_E_SHAPE, "linear_rows: unsupported M=%d N=%d K=%d act=%d", M, N, K, act);
    RowLnArgs g; memset(&g, 0, sizeof(g));
    g.A = x; g.lda = ldx; g.W = W; g.ldw = ldw; g.bias = b; g.add = add; g.ldadd = ldadd; g.mask = mask; g.ldmask = ldmask; g.Y = y; g.ldy = ldy;
    g.M = M; g.K = K; g.eps = 1e-5f;
    if (rowln_x6_enabled() && rowln_x6_ok(N, K, false)) return launch_rowln_x6(g, N, act, false, s);
    switch (act) {
        case ACT_RELU: return launch_rows_plain<ACT_RELU>(g, N, s);
        case ACT_SILU: return launch_rows_plain<ACT_SILU>(g, N, s);
        case ACT_GELU: return launch_rows_plain<ACT_GELU>(g, N, s);
        default: return launch_rows_plain<ACT_NONE>(g, N, s);
    }
}

bool linear_ln_act_rows_ok(int M, int N, int K, int act, const float* x, int ldx, const float* W, const float* add, int ldadd, const float* y, int ldy) {
    const bool al = ((((uintptr_t)x | (uintptr_t)W) & 15) == 0) && ldx % 4 == 0 && K % 4 == 0;
    return al && M >= 2048 && (N == 512 || N == 1024) && K >= 64 && (act == ACT_SILU || act == ACT_RELU || act == ACT_NONE);
}

// y[M,N] = act(LN(x W^T + b) * gamma + beta) + add, N in {512, 1024}; x [M,K] (row stride ldx), W [N,K] contiguous rows.
int linear_ln_act_rows(const float* x, int ldx, const float* W, const float* b, const float* gamma, const float* beta, int act, const float* add,
                       int ldadd, float* y, int ldy, int M, int N, int K, hipStream_t s, const float* mask, int ldmask) {
    AID_REQUIRE(linear_ln_act_rows_ok(M, N, K, act, x, ldx, W, add, ldadd, y, ldy), AID_E_SHAPE, "linear_ln_act_rows: unsupported M=%d N=%d K=%d act=%d", M, N, K, act);
    RowLnArgs g; memset(&g, 0, sizeof(g));
    g.A = x; g.lda = ldx; g.W = W; g.ldw = K; g.bias = b; g.gamma = gamma; g.beta = beta; g.add = add; g.ldadd = ldadd; g.mask = mask; g.ldmask = ldmask; g.Y = y; g.ldy = ldy;
    g.M = M; g.K = K; g.eps = 1e-5f;
    if (N == 1024 && g_rowln_split1024 && linear_rows_ok(M, N, K, ACT_NONE, x, ldx, W, K) && ldy == N) {
        // experiment (aid_rowln_debug(40)): the 1024-wide stage as two 512-column blocks of the plain full-row kernel (16-deep stages under the
        // stream hint: 74 KB of LDS, shares its CU) + a separate LayerNorm / act / skip pass over the 1024-wide rows
        AID_TRY(linear_rows(x, ldx, W, K, b, ACT_NONE, nullptr, 0, nullptr, 0, y, ldy, M, N, K, s));
        return ln_act(y, ldy, gamma, beta, act, mask, y, ldy, nullptr, M, N, s, add);
    }
    if (rowln_x6_enabled() && rowln_x6_ok(N, K, true)) {      // opt-in: the product as six bf16 MFMA terms (gemm_rowln_x6.hip)
        ProfScope prof(PROF_GEMM_TILE, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        prof.m = M; prof.n = N; prof.k = K; prof.variant = 28; prof.layout = act + 8 * (N == 1024);
        return launch_rowln_x6(g, N, act, true, s);
    }
    ProfScope prof(PROF_GEMM_TILE, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    prof.m = M; prof.n = N; prof.k = K; prof.variant = N == 1024 ? 20 : 21;
    const bool d16 = N == 512 && K % 16 == 0 && rowln_bkt16(s);
    prof.layout = act + 8 * (K % (N == 1024 || d16 ? 16 : 32) != 0) + 16 * (N == 1024 && K % 16 == 0 && g_rowln_dma) + 32 * d16;      // act, K-tail, DMA, 16-deep: the instantiation
    if (N == 1024) {
        if (act == ACT_SILU) return launch_rowln<1024, 16, ACT_SILU>(g, s);
        if (act == ACT_RELU) return launch_rowln<1024, 16, ACT_RELU>(g, s);
        return launch_rowln<1024, 16, ACT_NONE>(g, s);
    }
    if (act == ACT_SILU) return launch_rowln<512, 32, ACT_SILU>(g, s);
    if (act == ACT_RELU) return launch_rowln<512, 32, ACT_RELU>(g, s);
    return launch_rowln<512, 32, ACT_NONE>(g, s);
}

}  // namespace aid

using namespace aid;
extern "C" int aid_rowln_debug(int mode) { if (mode == 16) g_rowln_dma = 0; else if (mode == 17) g_rowln_dma = 1; else if (mode == 34) g_rowln_bkt16 = 1; else if (mode == 35) g_rowln_bkt16 = 0; else if (mode == 36) g_rowln_bkt16 = -1; else if (mode == 37) g_rowln_bkt8 = 1; else if (mode == 38) g_rowln_bkt8 = 0; else if (mode == 40) g_rowln_split1024 = 1; else if (mode == 41) g_rowln_split1024 = 0; else g_rowln_dbg = mode; return AID_OK; }
extern "C" int aid_linear_ln_act(const float* x, int ldx, const float* W, const float* b, const float* gamma, const float* beta, int act,
                                 const float* mask, const float* add, int ldadd, float* y, int ldy, int M, int N, int K, void* stream) {
    AID_REQUIRE(x && W && y, AID_E_NULL, "aid_linear_ln_act: null tensor");
    return linear_ln_act_rows(x, ldx, W, b, gamma, beta, act, add, ldadd, y, ldy, M, N, K, (hipStream_t)stream, mask, N);
}
