// Mid-M GEMM for the reverse-diffusion loop: C[M,N] = act(A W^T + b) (+ residual), or split-K partial slabs, at 129..1024 rows.
//
// Where it runs: the z-dependent trunk of one denoise step (reference: core/diffusion.py:176-206 -> models/score_networks.py:214-235,
// DiTBlock.forward) at the 256 / 512 / 768 rows a train_step's merged belief updates have.  Per block that is x Wov^T (512x512),
// x W1^T + GELU (512 -> 2048) and u W2^T (2048 -> 512): 0.1 - 1.6 GFLOP each, i.e. 1 - 10 us of the chip's fp32 matrix pipes, fifty
// times six times per loop.  The 64x64-tile kernels left the pipes mostly idle on these shapes: 384 tiles on 256 CUs (1.5 rounds),
// 96 tiles for the 512-wide outputs, every K tile behind a workgroup barrier with nothing else resident on the CU to fill it.
//
// MI355X mapping
//   * exactly one workgroup per CU: the row count is cut into 8 row tiles (BM = 32 ceil(M / 256): 96 rows at M = 768), columns into
//     64-wide tiles, and the K dimension of the 512-wide outputs into `ksplit` slices, so that a layer is 8 x 32 (x 1) or 8 x 8 x 4 = 256
//     equal work items.  Split-K slices write raw partial tiles to slabs [ksplit][M][N]; the NEXT kernel of the chain (the adaLN that
//     consumes the residual stream anyway, aid::adaln_sum) adds them up in a fixed order - no atomics, no extra launch, reproducible.
//   * 4 waves as 2 x 2, each (BM/2) x 32 outputs = BMT x 2 accumulators of v_mfma_f32_16x16x4_f32 (32-cycle issue, so a 96 x 64 x 64
//     stage is 3 072 cycles of each SIMD's pipe; 24 accumulator registers leave room for two stages of prefetch).
//   * K in stages of 64: [rows][64] LDS images with the 16-B chunk index XOR-ed by (row & 15) - b128 staging writes and b128 fragment
//     reads are bank-conflict free (checked exhaustively: tools/perf/README.md) - double buffered (80 KB at BM = 96).
//   * register staging with the hand-over hidden in the MFMA stream (the structure measured on gemm_rowln), prefetch distance 2: the 6 + 4
//     global loads of tile kt + 2 are issued one by one between the MFMAs of the first half of stage kt, the LDS writes of tile kt + 1
//     between those of the second half, the fragments of k-group g + 1 are read while group g is contracted; one barrier per stage.
//   * a lane's float4 fragment holds k = 16 g + 4 (lane >> 4) + {0..3} of its row; MFMA j of the group contracts element j of every
//     lane, i.e. k in {16 g + j, + 4, + 8, + 12}: the same permutation for both operands, so the reduction order is fixed and k-contiguous
//     operands need no transpose.
//   * blockIdx -> tile: the 8 row tiles of one (column tile, K slice) sit on ONE XCD (b % 8 labels the XCD group; speed only), so a
//     weight panel is pulled from the Infinity Cache once per layer and the 1.5 MB activation operand once per XCD.
#include "aid_common.h"
#include "rowwise.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gemm_mid.hip: the in-kernel split-K fold relies on gfx942 / gfx950 sc1 (agent-scope, L2-bypassing) store-acknowledge semantics; build with --offload-arch=gfx950"
#endif

namespace aid {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MidArgs {
    const float* A; int lda;              // [M][K]
    const float* W; int ldw;              // [N][K], or [K][N] when the kernel is instantiated with BKM (dX = dY W: the weight as stored)
    const float* bias;                    // [N] or null (ignored in slab mode)
    const float* residual; int ldr;       // optional, added after the activation (ksplit == 1 only)
    float* C; int ldc;                    // ksplit == 1: output; else unused
    float* preact;                        // optional second output (row stride ldc): the value before the activation (training forward)
    const float* dact_pre; int dact;      // optional (row stride ldc): multiply by act'(dact_pre) of kind `dact` (tangent pass through a saved pre-activation)
    float* slab;                          // ksplit > 1: [ksplit][M][N] raw partial products
    int* ctr;                             // ksplit > 1, optional: one arrival counter per output tile (zero at launch, zero again at exit).  The
                                          //   workgroup that arrives LAST at a tile folds the slices in slice order and runs the full epilogue
                                          //   into C: a split-K GEMM that is complete when the kernel ends, bit-reproducible (gemm_f32's route)
    int accumulate;                       // C += (act must be NONE; ksplit == 1 or folded)
    int M, N, K, ksplit, act;
};

// issue priority of the reverse loop's kernels (see gemm_mid_kernel_o3); -DAID_CHAIN_PRIO_LEVEL=0 builds them without it (A/B)
#ifndef AID_CHAIN_PRIO_LEVEL
#define AID_CHAIN_PRIO_LEVEL 3
#endif
#define AID_CHAIN_PRIO() do { if (AID_CHAIN_PRIO_LEVEL > 0) __builtin_amdgcn_s_setprio(AID_CHAIN_PRIO_LEVEL); } while (0)
constexpr int MID_BK = 64, MID_BN = 64;
__device__ __forceinline__ int mid_off(int row, int chunk) { return row * MID_BK + ((chunk ^ (row & 15)) << 2); }

template <int ACT> __device__ __forceinline__ float mid_act(float x) {
    if constexpr (ACT == ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    else if constexpr (ACT == ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + expf(-x));
    else if constexpr (ACT == ACT_RELU) return x > 0.f ? x : 0.f;
    else return x;
}

// Work decomposition: 4 waves as WM x WN; a wave owns BMF x NJ accumulator blocks of 16 x 16: BM = 16 BMF WM rows, BN = 16 NJ WN columns.
//   <BMT, 2, 2, 2>: the reverse loop's 32 BMT x 64 tiles (8 row tiles at 129..1024 rows, 128-row tiles beyond)
//   <5, 1, 4, 1>  : 80 x 64 tiles - the rollout / BPTT layers at K*B = 2 560 rows x 512 columns are then exactly 32 x 8 = 256 workgroups, one
//                   per CU (the 64 x 64 tiling is 320 tiles: a quarter of the CUs run two)
// BKM: the W operand is contraction-major ([K][N], n contiguous): the data gradient dX = dY W reads the weight as stored, without a
// transposed copy.  Its stage is kept in LDS as it arrives ([64][BN + 4]: 16-B writes) and fragments are gathered with four 4-B reads per
// k-group (16 consecutive n at one k: conflict-free), in the shadow of the MFMAs.
template <int BMF, int WM, int WN, int NJ, int ACT, bool BKM>
__device__ __forceinline__ void gemm_mid_body(const MidArgs& g, int tiles_m, int tiles_n) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int NT = 256, BM = 16 * BMF * WM, BN = 16 * NJ * WN, BK = MID_BK, CPR = BK / 4;
    constexpr int BLD = BN + 4;                                                   // row pitch of the contraction-major W image
    constexpr int A_FL = BM * BK, B_FL = BKM ? BK * BLD : BN * BK;
    constexpr int PA = BM * CPR / NT, PB = BN * CPR / NT, NS = PA + PB;          // float4 pieces per thread per stage
    static_assert(PA * NT == BM * CPR && PB * NT == BN * CPR, "tile does not split into whole pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][BK]
    float* Bs = smem + 2 * A_FL;      // [2][BN][BK]  or  [2][BK][BLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wm = (wave / WN) * (BM / WM), wn = (wave % WN) * (BN / WN);
    // tile map: T = (column tile, K slice) pairs; whole rounds of 8 pairs go one per XCD group, each with its `tiles_m` row tiles
    int tm, tn, ky;
    {
        const int b = blockIdx.x, npair = tiles_n * g.ksplit, n8 = npair & ~7, full = n8 * tiles_m;
        int T;
        if (b < full) { const int x = b & 7, q = b >> 3; T = x + 8 * (q / tiles_m); tm = q % tiles_m; }
        else { const int r = b - full; T = n8 + r / tiles_m; tm = r % tiles_m; }
        // K slice fastest: an XCD group (T % 8) then sees ONE K slice when ksplit divides 8, i.e. 1/ksplit of the activation operand instead
        // of all of it (768 x 2048 layer: 17 MB instead of 54 MB pulled through the eight L2s per launch, rocprofv3 FETCH_SIZE)
        ky = T % g.ksplit; tn = T / g.ksplit;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kper = g.K / g.ksplit, kbeg = ky * kper, nk = kper / BK;
    f32x4 acc[BMF][NJ];
#pragma unroll
    for (int i = 0; i < BMF; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging maps: piece p < PA is an A piece, else a W piece; thread f = tid + NT p covers (row f / 16, chunk f % 16) of a k-contiguous
    // operand, (k f / (BN / 4), column chunk f % (BN / 4)) of the contraction-major W
    const float* src[NS]; int dst[NS];
    const size_t wstep = BKM ? (size_t)BK * g.ldw : (size_t)BK;                   // one K stage further along W
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        const int f = tid + NT * (p < PA ? p : p - PA), row = f / CPR, c = f % CPR;
        if (p < PA) { src[p] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + kbeg + 4 * c; dst[p] = mid_off(row, c); }
        else if (!BKM) { src[p] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + kbeg + 4 * c; dst[p] = 2 * A_FL + mid_off(row, c); }
        else { const int kk = f / (BN / 4), cq = f % (BN / 4); src[p] = g.W + (size_t)(kbeg + kk) * g.ldw + n0 + 4 * cq; dst[p] = 2 * A_FL + kk * BLD + 4 * cq; }
    }
    // Prefetch distance 2: while tile kt is contracted, tile kt + 2 is loaded into the register set tile kt + 1 does not use and tile kt + 1
    // (loaded a whole stage ago) is written to the other LDS buffer: a global load has 1.5 stages (~2 us) to arrive - the weights of a block
    // are cold every denoise step (56 MB stream through 32 MB of L2) and one wave per SIMD has nobody to hide a late load behind.  With one
    // workgroup per CU registers are free (122 + 40 of 512).  The loop is unrolled by two so both sets are named statically.
    f32x4 rg0[NS], rg1[NS];
    auto load_piece = [&](f32x4 (&rg)[NS], int p, int kt) { rg[p] = *reinterpret_cast<const f32x4*>(src[p] + (p < PA ? (size_t)kt * BK : (size_t)kt * wstep)); };
    auto store_piece = [&](const f32x4 (&rg)[NS], int p, int buf) { *reinterpret_cast<f32x4*>(smem + dst[p] + buf * (p < PA ? A_FL : B_FL)) = rg[p]; };
    constexpr int NG = BK / 16, MPG = 4 * BMF * NJ, NM = NG * MPG, S0 = NM / 2;
    constexpr int LSTEP = S0 / NS > 0 ? S0 / NS : 1, SSTEP = (NM - S0) / NS > 0 ? (NM - S0) / NS : 1;
    static_assert(S0 / NS >= 1, "more staging pieces than MFMA slots in half a stage");
    // fragments of k-group gk: a lane's four values are k = 16 gk + 4 kq + {0..3} of its row / column
    auto read_a = [&](const float* a_s, int gk, f32x4 (&av)[BMF]) {
#pragma unroll
        for (int i = 0; i < BMF; ++i) av[i] = *reinterpret_cast<const f32x4*>(a_s + 16 * i * BK + (((4 * gk + kq) ^ r16) << 2));
    };
    auto read_b = [&](const float* b_s, int gk, f32x4 (&bv)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (!BKM) bv[j] = *reinterpret_cast<const f32x4*>(b_s + 16 * j * BK + (((4 * gk + kq) ^ r16) << 2));
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = b_s[(16 * gk + 4 * kq + e) * BLD + 16 * j];
            }
        }
    };
    // contract LDS buffer `cur`; first half: load tile `kt_load` into `ld`; second half: write `st` (the tile after this one) to buffer cur ^ 1
    auto stage = [&](int cur, int kt_load, f32x4 (&ld)[NS], const f32x4 (&st)[NS]) {
        const float* a_s = As + cur * A_FL + (wm + r16) * BK;
        // (row & 15) == r16 for every fragment row of this lane: the swizzle term is lane-constant
        const float* b_s = BKM ? Bs + cur * B_FL + wn + r16 : Bs + cur * B_FL + (wn + r16) * BK;
        f32x4 av[2][BMF], bv[2][NJ];
        read_a(a_s, 0, av[0]); read_b(b_s, 0, bv[0]);
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) {
            const int pc = gk & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < BMF; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int idx = ((gk * 4 + q) * BMF + i) * NJ + j;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[pc][i][q], bv[pc][j][q], acc[i][j], 0, 0, 0);
                        if (idx % MPG == (MPG > 1 ? 1 : 0) && gk + 1 < NG) { read_a(a_s, gk + 1, av[pc ^ 1]); read_b(b_s, gk + 1, bv[pc ^ 1]); }   // next k-group's fragments, early in this group
                        if (idx < S0 && idx % LSTEP == 0 && idx / LSTEP < NS) { load_piece(ld, idx / LSTEP, kt_load); __builtin_amdgcn_sched_barrier(0); }
                        if (idx >= S0 && (idx - S0) % SSTEP == 0 && (idx - S0) / SSTEP < NS) { store_piece(st, (idx - S0) / SSTEP, cur ^ 1); __builtin_amdgcn_sched_barrier(0); }
                    }
        }
    };
    // prologue: tile 0 -> LDS buffer 0, tile 1 -> register set 1 (written to buffer 1 during stage 0)
    const int last = nk - 1;
#pragma unroll
    for (int p = 0; p < NS; ++p) load_piece(rg0, p, 0);
#pragma unroll
    for (int p = 0; p < NS; ++p) load_piece(rg1, p, min(1, last));
#pragma unroll
    for (int p = 0; p < NS; ++p) store_piece(rg0, p, 0);
    __syncthreads();
    // nk is even (checked on the host).  Tiles beyond the end are clamped to the last one: they are loaded / written to an idle buffer and
    // never contracted, which keeps the loop branch-free.
    for (int kt = 0; kt < nk; kt += 2) {
        stage(0, min(kt + 2, last), rg0, rg1);       // contract tile kt (buffer 0); load kt + 2 -> set 0; write tile kt + 1 (set 1) -> buffer 1
        __syncthreads();
        stage(1, min(kt + 3, last), rg1, rg0);       // contract tile kt + 1 (buffer 1); load kt + 3 -> set 1; write tile kt + 2 (set 0) -> buffer 0
        __syncthreads();
    }
    // ---- epilogue.  C/D of 16x16x4: column = lane & 15, rows 4 (lane >> 4) + {0..3}
    if (g.ksplit > 1) {
        const bool fold = g.ctr != nullptr;
#pragma unroll
        for (int i = 0; i < BMF; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = n0 + wn + 16 * j + r16;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m0 + wm + 16 * i + 4 * kq + e;
                    if (row < g.M && col < g.N) {
                        float* p = g.slab + ((size_t)ky * g.M + row) * g.N + col;
                        if (fold) __hip_atomic_store(p, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // written through the XCD's L2
                        else *p = acc[i][j][e];
                    }
                }
            }
        if (!fold) return;                // raw slabs: the consumer folds them
        // The slices of a tile may sit on different XCDs (one L2 each, not coherent with each other inside a kernel): partials are stored
        // and read back with device-scope accesses, which pass the L2 (no cache write-back / invalidate: an agent-scope fence per workgroup
        // costs ~40 us here), a thread waits for its own stores to be acknowledged before the barrier, and one thread then counts the
        // arrival.  Whoever counts the last one reads the other slices and finishes the tile.  The fold runs in slice order whoever folds,
        // and a slice's partial is the same bits from a register or from memory: the result does not depend on the arrival order.
        //
        // Ordering contract (hardware, not the C++ memory model - the atomics below are RELAXED on purpose): on gfx942 / gfx950 a
        // `global_store ... sc1` is performed at agent scope (written through the XCD's non-coherent L2 to the memory side every XCD sees)
        // and its vmcnt slot is released only when that write is acknowledged there; `s_waitcnt vmcnt(0)` therefore means "my partials are
        // visible device-wide", the workgroup barrier extends that to all 256 threads, and the arrival count issued after it cannot be
        // observed before them.  On the reading side `global_load ... sc1` misses the local L2 by scope, and the loads are issued only after
        // the thread that counted the last arrival has published s_last through LDS + barrier.  No other target is compiled (see the
        // #error below); tests/test_hip_kernels.py::test_gemm_split_k_folded_in_kernel poisons the slices with NaN between launches and
        // changes the operands every launch, so a stale or early read shows up as NaN / a wrong sum.  If it ever trips: make the counter
        // increment __ATOMIC_RELEASE and the fold's first load __ATOMIC_ACQUIRE (cost: the L2 write-back measured above).
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tile = tn * tiles_m + tm;
        if (tid == 0) s_last = __hip_atomic_fetch_add(g.ctr + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.ksplit - 1;
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) __hip_atomic_store(g.ctr + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // for the next launch on this stream
        f32x4 sum[BMF][NJ];
        for (int y0 = 0; y0 < g.ksplit; y0 += 4) {    // slice order; four slices' loads are in flight together
            f32x4 part[4][BMF][NJ];
#pragma unroll
            for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                for (int i = 0; i < BMF; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int col = n0 + wn + 16 * j + r16, y = y0 + yy;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = m0 + wm + 16 * i + 4 * kq + e;
                            part[yy][i][j][e] = acc[i][j][e];
                            if (y != ky && y < g.ksplit && row < g.M && col < g.N)
                                part[yy][i][j][e] = __hip_atomic_load(g.slab + ((size_t)y * g.M + row) * g.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
            for (int yy = 0; yy < 4; ++yy)
                if (y0 + yy < g.ksplit) {
#pragma unroll
                    for (int i = 0; i < BMF; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) sum[i][j] = y0 + yy == 0 ? part[yy][i][j] : sum[i][j] + part[yy][i][j];
                }
        }
#pragma unroll
        for (int i = 0; i < BMF; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = sum[i][j];
    }
#pragma unroll
    for (int i = 0; i < BMF; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn + 16 * j + r16;
            const float bcol = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm + 16 * i + 4 * kq + e;
                if (row < g.M && col < g.N) {
                    float v = acc[i][j][e] + bcol;
                    if (g.preact) g.preact[(size_t)row * g.ldc + col] = v;
                    v = mid_act<ACT>(v);
                    if (g.dact_pre) v *= act_grad(g.dact_pre[(size_t)row * g.ldc + col], g.dact);
                    if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                    if (g.accumulate) v += g.C[(size_t)row * g.ldc + col];
                    g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
}

template <int BMF, int WM, int WN, int NJ, int ACT, bool BKM>
__global__ void __launch_bounds__(256) gemm_mid_kernel(MidArgs g, int tiles_m, int tiles_n) { gemm_mid_body<BMF, WM, WN, NJ, ACT, BKM>(g, tiles_m, tiles_n); }
// The 32-row tile (129..256 rows: the three concurrent 256-row belief chains of a train_step) with its registers capped at 512 / 3: its
// workgroup is 48 KB of LDS, so three fit a CU - but at the 184 VGPRs the compiler takes by itself only two waves fit a SIMD and the third
// chain's workgroup waits for a slot (three chains overlapped 1.63x, exactly what two resident workgroups give).
template <int BMF, int WM, int WN, int NJ, int ACT, bool BKM>
__global__ void __launch_bounds__(256, 3) gemm_mid_kernel_o3(MidArgs g, int tiles_m, int tiles_n) {
    // (round 5) The reverse loop's kernels raise their waves' issue priority: a belief chain runs on a side stream UNDERNEATH the estimator's tall full-row
    // kernels, and where a chain workgroup is co-resident with one (the 512-wide stages) its few short waves were losing the SIMD's issue arbitration to the
    // tall kernel's two older, MFMA-dense waves (age decides at equal priority; priority outranks age: MI355X_MICROARCH.md, "Two waves per SIMD").  The chain is
    // latency-bound - every cycle its wave waits for an issue slot is a cycle of the step - while the tall kernel barely notices a 7-us kernel's instructions.
    // cfg-2 step with the dead belief elided 44.27 -> 43.22 ms, as executed 48.99 -> 48.50 (two alternating pairs, tools/perf/ab_bench.sh).  No arithmetic changes.
    AID_CHAIN_PRIO();
    gemm_mid_body<BMF, WM, WN, NJ, ACT, BKM>(g, tiles_m, tiles_n);
}

template <int BMF, int WM, int WN, int NJ, int ACT, bool BKM>
static int launch_mid(const MidArgs& g, hipStream_t s) {
    constexpr int BM = 16 * BMF * WM, BN = 16 * NJ * WN;
    constexpr size_t lds = sizeof(float) * 2 * ((size_t)BM * MID_BK + (BKM ? (size_t)MID_BK * (BN + 4) : (size_t)BN * MID_BK));
    static bool attr_done = false;
    if (lds > 65536 && !attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_mid_kernel<BMF, WM, WN, NJ, ACT, BKM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    if constexpr (BMF == 1 && WM == 2) hipLaunchKernelGGL((gemm_mid_kernel_o3<BMF, WM, WN, NJ, ACT, BKM>), dim3(tiles_m * tiles_n * g.ksplit), dim3(256), lds, s, g, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_mid_kernel<BMF, WM, WN, NJ, ACT, BKM>), dim3(tiles_m * tiles_n * g.ksplit), dim3(256), lds, s, g, tiles_m, tiles_n);
    AID_LAUNCH_CHECK();
    return AID_OK;
}
// 80-row tiles when they make the layer (close to) one workgroup per CU: M a multiple of 80 beyond the 8-row-tile regime
static inline bool mid_tall80(int M, int N, int ksplit) {
    if (M <= 1024 || M % 80 != 0 || N % MID_BN != 0) return false;
    const long wg = (long)(M / 80) * (N / MID_BN) * ksplit;
    return wg >= 192 && wg <= 512;      // (2 560 x 256 as 128 workgroups was measured: slower than the 64 x 64 tiling, 5.41 against 5.23 ms per rollout + BPTT)
}
static int g_mid_bmt_min = 1, g_mid_bmt_max = 4;
template <int ACT, bool BKM>
static int launch_mid_bm(const MidArgs& g, hipStream_t s) {
    if (mid_tall80(g.M, g.N, g.ksplit)) return launch_mid<5, 1, 4, 1, ACT, BKM>(g, s);
    int bmt = (g.M + 255) / 256;        // up to 1024 rows: 8 row tiles of BM = 32 ceil(M / 256); beyond: ceil(M / 128) tiles of 128 rows
    if (bmt < g_mid_bmt_min) bmt = g_mid_bmt_min;      // timing experiments (aid_mid_debug)
    if (bmt > g_mid_bmt_max) bmt = g_mid_bmt_max;
    switch (bmt) {
        case 1: return launch_mid<1, 2, 2, 2, ACT, BKM>(g, s);
        case 2: return launch_mid<2, 2, 2, 2, ACT, BKM>(g, s);
        case 3: return launch_mid<3, 2, 2, 2, ACT, BKM>(g, s);
        default: return launch_mid<4, 2, 2, 2, ACT, BKM>(g, s);
    }
}

bool gemm_mid_shape_ok(int M, int N, int K, int ksplit, int act, const float* A, int lda, const float* W, int ldw) {
    const bool al = ((((uintptr_t)A | (uintptr_t)W) & 15) == 0) && lda % 4 == 0 && ldw % 4 == 0;
    return al && M >= 1 && M <= MID_MAX_M && N >= 64 && ksplit >= 1 && K % (2 * MID_BK * ksplit) == 0 &&      // an even number of 64-deep stages per slice
           (act == ACT_NONE || act == ACT_GELU || act == ACT_SILU || act == ACT_RELU);
}
bool gemm_mid_ok(int M, int N, int K, int ksplit, int act, const float* A, int lda, const float* W, int ldw) {
    return M > 128 && gemm_mid_shape_ok(M, N, K, ksplit, act, A, lda, W, ldw);
}

// ksplit == 1: C = act(A W^T + bias) + residual (preact, optional: the value before the activation).  ksplit > 1: slab[y] = A[:, Ky] W[:, Ky]^T
// (raw partials; bias / act / residual are the consumer's: adaln_sum or slab_sum; act must be NONE then) - or, with `ctr` (one zeroed int per
// output tile), the slices are folded in the kernel and C gets the full epilogue as at ksplit == 1.
// w_kmajor: W is [K][N] (row stride ldw) - the data-gradient form; needs N % 64 == 0.
int gemm_mid_ex(const float* A, int lda, const float* W, int ldw, int w_kmajor, const float* bias, int act, const float* residual, int ldr, float* C, int ldc,
                float* preact, float* slab, int M, int N, int K, int ksplit, hipStream_t s, const float* dact_pre, int dact, int* ctr, int accumulate) {
    AID_REQUIRE(gemm_mid_shape_ok(M, N, K, ksplit, act, A, lda, W, ldw), AID_E_SHAPE, "gemm_mid: unsupported M=%d N=%d K=%d ksplit=%d act=%d", M, N, K, ksplit, act);
    AID_REQUIRE(ksplit == 1 ? C != nullptr : (slab != nullptr && (ctr ? C != nullptr : (act == ACT_NONE && !residual && !preact && !accumulate))), AID_E_SHAPE,
                "gemm_mid: split-K writes raw slabs unless tile counters are given");
    AID_REQUIRE(!accumulate || act == ACT_NONE, AID_E_SHAPE, "gemm_mid: accumulate with an activation");
    AID_REQUIRE(!w_kmajor || (N % MID_BN == 0 && act == ACT_NONE), AID_E_SHAPE, "gemm_mid: contraction-major W needs N %% 64 == 0 and no activation (N=%d)", N);
    MidArgs g; memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.preact = preact; g.slab = slab;
    g.M = M; g.N = N; g.K = K; g.ksplit = ksplit; g.act = act; g.dact_pre = (ksplit == 1 || ctr) ? dact_pre : nullptr; g.dact = dact;
    g.ctr = ksplit > 1 ? ctr : nullptr; g.accumulate = accumulate;
    ProfScope prof(PROF_GEMM, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const int bmt = mid_tall80(M, N, ksplit) ? 5 : ((M + 255) / 256 > 4 ? 4 : (M + 255) / 256);
    prof.m = M; prof.n = N; prof.k = K; prof.layout = 8 * bmt + act + (w_kmajor ? 64 : 0); prof.variant = g.ctr ? 24 : 22;      // layout field: 8 BMF + act (+ 64: contraction-major W) = the instantiation
    if (w_kmajor) return launch_mid_bm<ACT_NONE, true>(g, s);
    switch (act) {
        case ACT_GELU: return launch_mid_bm<ACT_GELU, false>(g, s);
        case ACT_SILU: return launch_mid_bm<ACT_SILU, false>(g, s);
        case ACT_RELU: return launch_mid_bm<ACT_RELU, false>(g, s);
        default: return launch_mid_bm<ACT_NONE, false>(g, s);
    }
}
int gemm_mid(const float* A, int lda, const float* W, int ldw, const float* bias, int act, const float* residual, int ldr, float* C, int ldc,
             float* slab, int M, int N, int K, int ksplit, hipStream_t s) {
    return gemm_mid_ex(A, lda, W, ldw, 0, bias, act, residual, ldr, C, ldc, nullptr, slab, M, N, K, ksplit, s, nullptr, 0, nullptr, 0);
}

// ---- <= 128 rows: a 32 x 16 output tile per workgroup, its four waves a quarter of the K slice each ---------------------------------------
// act() / collector batches and the 32-row shards of a strong-scaling run (reference: agents/state_agent.py:72-108, utils/async_collector.py:530-595).
// At these row counts a layer is 0.03 - 0.13 GFLOP: nothing but the latency chain of its kernel matters, and that chain is shortest when there is
// no pipeline at all: every lane requests ALL its operand fragments at once (KW / 16 float4 loads per 16-row / 16-column block, straight from
// global memory into the registers v_mfma_f32_16x16x4_f32 reads - no LDS staging, no stage barriers), the wave runs its KW / 4 MFMAs per
// block, the NW (four or eight) partial tiles meet in LDS and are added in wave order, and the epilogue (bias, activation) or the raw K-slice slab
// is written once.  16-column tiles make N / 16 workgroups per row tile: 128 for the 2048-wide layer, 32 x ksplit for the 512-wide ones.
// Fragment layout as in gemm_mid_body: a lane's float4 holds k = 16 g + 4 (lane >> 4) + {0..3} of row / column (lane & 15); MFMA j of group g
// contracts element j of every lane - the same k permutation on both operands.
template <int KW, int NW, int ACT>
__global__ void __launch_bounds__(64 * NW) gemm_w4_kernel(MidArgs g, int tiles_m) {
    constexpr int NG = KW / 16;
    __shared__ __attribute__((aligned(16))) float red[NW][2][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x;
    const int tm = b % tiles_m, rest = b / tiles_m, ky = rest % g.ksplit, tn = rest / g.ksplit;
    const int m0 = tm * 32, n0 = tn * 16;
    const int kbeg = ky * (g.K / g.ksplit) + wave * KW + 4 * kq;
    const float* a0 = g.A + (size_t)min(m0 + r16, g.M - 1) * g.lda + kbeg;
    const float* a1 = g.A + (size_t)min(m0 + 16 + r16, g.M - 1) * g.lda + kbeg;
    const float* w0 = g.W + (size_t)min(n0 + r16, g.N - 1) * g.ldw + kbeg;
    f32x4 av0[NG], av1[NG], bv[NG];
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) {
        bv[gk] = *reinterpret_cast<const f32x4*>(w0 + 16 * gk);
        av0[gk] = *reinterpret_cast<const f32x4*>(a0 + 16 * gk);
        av1[gk] = *reinterpret_cast<const f32x4*>(a1 + 16 * gk);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gk = 0; gk < NG; ++gk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[gk][q], bv[gk][q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[gk][q], bv[gk][q], acc1, 0, 0, 0);
        }
    *reinterpret_cast<f32x4*>(&red[wave][0][lane][0]) = acc0;
    *reinterpret_cast<f32x4*>(&red[wave][1][lane][0]) = acc1;
    __syncthreads();
    // 2 blocks x 64 lanes x 4 = 512 outputs over the first 256 threads, two each: (block i, lane l, element pair); C/D of 16x16x4: column = l & 15,
    // row = 4 (l >> 4) + e
    if (tid >= 256) return;
    const int i = tid >> 7, l = (tid >> 1) & 63, e0 = 2 * (tid & 1);
#pragma unroll
    for (int e = e0; e < e0 + 2; ++e) {
        float v = red[0][i][l][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][i][l][e];                          // fixed order: wave 0, 1, 2, ...
        const int row = m0 + 16 * i + 4 * (l >> 4) + e, col = n0 + (l & 15);
        if (row < g.M && col < g.N) {
            if (g.ksplit > 1) g.slab[((size_t)ky * g.M + row) * g.N + col] = v;
            else g.C[(size_t)row * g.ldc + col] = mid_act<ACT>(v + (g.bias ? g.bias[col] : 0.f));
        }
    }
}
// (round 5) The same kernel with the adaLN that produces its A operand as a PROLOGUE, for <= 32 rows (act(), the collector's smallest batches):
//   x_new = res + bias_in + sum_y slab_in[y] (slab order);  A = LN(x_new) (1 + scale) + shift     (exactly adaln_sum_kernel's arithmetic and order)
// Every workgroup rebuilds the (at most 32 x 512) A tile into LDS - a wave per row, all loads of a row in flight at once - while its weight fragments are
// already on their way; the workgroup of column tile 0 / K slice 0 also writes x_new, the residual stream.  At these row counts a denoise step is a chain of
// 32 dependent launches of ~3.3 us each whatever they compute; folding the 12 adaLN launches into their consumers leaves 20.  (At 32+ rows per tile the
// redundant prologue costs more than the launch it saves - round 3 measured that with the skinny kernel - so the caller uses this form up to W4LN_MAX_ROWS.)
// The input slabs and the output slabs must be different buffers (other workgroups may still be reading the former when this one writes the latter).
struct W4Ln {
    const float* res; int ldres;                 // [M][D]
    const float* slab; size_t slab_stride;       // nslab raw K-slice slabs [M][D], or unused with NSLAB == 0
    const float* bias;                           // [D] or null
    const float* mod; int ldmod;                 // row m: [scale (D) | shift (D)]
    float* xout; int ldxo;                       // x_new, written once (may be null)
};
constexpr int W4LN_D = 512, W4LN_LD = W4LN_D + 4, W4LN_MAX_ROWS = 16;
template <int KW, int NW, int ACT, int NSLAB>
__global__ void __launch_bounds__(64 * NW) gemm_w4ln_kernel(MidArgs g, W4Ln ln) {
    constexpr int NG = KW / 16, D = W4LN_D, NV = D / 256;
    extern __shared__ __attribute__((aligned(16))) float w4smem[];
    float* ys = w4smem;                                               // [32][W4LN_LD]: the normalised A tile (rows >= M are never read: row indices are clamped)
    float (*red)[2][64][4] = reinterpret_cast<float (*)[2][64][4]>(w4smem + 32 * W4LN_LD);      // [NW][2][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x;
    const int ky = b % g.ksplit, tn = b / g.ksplit;                   // one row tile (M <= 32)
    const int n0 = tn * 16;
    const int kloc = ky * (g.K / g.ksplit) + wave * KW + 4 * kq;      // this lane's first k
    // the weight fragments do not depend on the prologue: request them first
    const float* w0 = g.W + (size_t)min(n0 + r16, g.N - 1) * g.ldw + kloc;
    f32x4 bv[NG];
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) bv[gk] = *reinterpret_cast<const f32x4*>(w0 + 16 * gk);
    // ---- prologue: rows wave, wave + NW, ... (adaln_sum_kernel's row body)
    const bool writer = ln.xout != nullptr && tn == 0 && ky == 0;
    for (int row = wave; row < g.M; row += NW) {
        const f32x4* sc = reinterpret_cast<const f32x4*>(ln.mod + (size_t)row * ln.ldmod);
        const f32x4* sh = reinterpret_cast<const f32x4*>(ln.mod + (size_t)row * ln.ldmod + D);
        f32x4 v[NV], a[NV], bsh[NV], bi[NV], sl[NSLAB > 0 ? NSLAB : 1][NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = reinterpret_cast<const f32x4*>(ln.res + (size_t)row * ln.ldres)[lane + 64 * i];
            a[i] = sc[lane + 64 * i]; bsh[i] = sh[lane + 64 * i];
            bi[i] = ln.bias ? reinterpret_cast<const f32x4*>(ln.bias)[lane + 64 * i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int y0 = 0; y0 < NSLAB; ++y0)
#pragma unroll
            for (int i = 0; i < NV; ++i) sl[y0][i] = reinterpret_cast<const f32x4*>(ln.slab + (size_t)y0 * ln.slab_stride + (size_t)row * D)[lane + 64 * i];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] += bi[i];
#pragma unroll
            for (int y0 = 0; y0 < NSLAB; ++y0) v[i] += sl[y0][i];
        }
        if (writer) {
#pragma unroll
            for (int i = 0; i < NV; ++i) reinterpret_cast<f32x4*>(ln.xout + (size_t)row * ln.ldxo)[lane + 64 * i] = v[i];
        }
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = wave_sum(sm) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * (1.0f + a[i][j]) + bsh[i][j];
            *reinterpret_cast<f32x4*>(ys + row * W4LN_LD + 4 * (lane + 64 * i)) = o;
        }
    }
    __syncthreads();
    // ---- the product, A fragments from LDS (row pitch D + 4: the 16 rows of a fragment read hit distinct banks)
    const float* a0 = ys + min(r16, g.M - 1) * W4LN_LD + kloc;
    const float* a1 = ys + min(16 + r16, g.M - 1) * W4LN_LD + kloc;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) {
        const f32x4 av0 = *reinterpret_cast<const f32x4*>(a0 + 16 * gk), av1 = *reinterpret_cast<const f32x4*>(a1 + 16 * gk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[q], bv[gk][q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[q], bv[gk][q], acc1, 0, 0, 0);
        }
    }
    *reinterpret_cast<f32x4*>(&red[wave][0][lane][0]) = acc0;
    *reinterpret_cast<f32x4*>(&red[wave][1][lane][0]) = acc1;
    __syncthreads();
    if (tid >= 256) return;
    const int i = tid >> 7, l = (tid >> 1) & 63, e0 = 2 * (tid & 1);
#pragma unroll
    for (int e = e0; e < e0 + 2; ++e) {
        float v = red[0][i][l][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][i][l][e];                          // fixed order: wave 0, 1, 2, ... (as gemm_w4_kernel)
        const int row = 16 * i + 4 * (l >> 4) + e, col = n0 + (l & 15);
        if (row < g.M && col < g.N) {
            if (g.ksplit > 1) g.slab[((size_t)ky * g.M + row) * g.N + col] = v;
            else g.C[(size_t)row * g.ldc + col] = mid_act<ACT>(v + (g.bias ? g.bias[col] : 0.f));
        }
    }
}
// waves per workgroup x K per wave for a K slice of `kper`: 64 per wave (12 loads in flight per lane) wherever four or eight waves cover the slice, else 128
static inline bool w4_shape(int kper, int& nw, int& kw) {
    if (kper == 512) { nw = 8; kw = 64; return true; }
    if (kper == 256) { nw = 4; kw = 64; return true; }
    if (kper == 1024) { nw = 8; kw = 128; return true; }
    return false;
}
bool gemm_w4_ok(int M, int N, int K, int ksplit, int act, const float* A, int lda, const float* W, int ldw) {
    const bool al = ((((uintptr_t)A | (uintptr_t)W) & 15) == 0) && lda % 4 == 0 && ldw % 4 == 0;
    int nw, kw;
    return al && M >= 1 && M <= 1024 && N >= 16 && ksplit >= 1 && K % ksplit == 0 && w4_shape(K / ksplit, nw, kw) && (ksplit == 1 || act == ACT_NONE) &&
           (act == ACT_NONE || act == ACT_GELU || act == ACT_SILU);
}
// ksplit == 1: C = act(A W^T + bias);  ksplit > 1: slab[y] = A[:, Ky] W[:, Ky]^T (raw).  K / ksplit in {256, 512, 1024}.
int gemm_w4(const float* A, int lda, const float* W, int ldw, const float* bias, int act, float* C, int ldc, float* slab, int M, int N, int K, int ksplit,
            hipStream_t s) {
    AID_REQUIRE(gemm_w4_ok(M, N, K, ksplit, act, A, lda, W, ldw) && (ksplit == 1 ? C != nullptr : slab != nullptr), AID_E_SHAPE,
                "gemm_w4: unsupported M=%d N=%d K=%d ksplit=%d act=%d", M, N, K, ksplit, act);
    MidArgs g; memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.slab = slab; g.M = M; g.N = N; g.K = K; g.ksplit = ksplit; g.act = act;
    ProfScope prof(PROF_GEMM, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)ksplit * M * N));
    prof.m = M; prof.n = N; prof.k = K; prof.layout = act; prof.variant = 26;
    int nw = 0, kw = 0; w4_shape(K / ksplit, nw, kw);
    const int tiles_m = (M + 31) / 32;
    const dim3 grid(tiles_m * ((N + 15) / 16) * ksplit);
#define AID_W4(KW_, NW_, ACT_) hipLaunchKernelGGL((gemm_w4_kernel<KW_, NW_, ACT_>), grid, dim3(64 * NW_), 0, s, g, tiles_m)
#define AID_W4A(KW_, NW_) { if (act == ACT_GELU) AID_W4(KW_, NW_, ACT_GELU); else if (act == ACT_SILU) AID_W4(KW_, NW_, ACT_SILU); else AID_W4(KW_, NW_, ACT_NONE); }
    if (kw == 64 && nw == 8) AID_W4A(64, 8) else if (kw == 64) AID_W4A(64, 4) else AID_W4A(128, 8)
#undef AID_W4A
#undef AID_W4
    AID_LAUNCH_CHECK();
    return AID_OK;
}

bool gemm_w4ln_ok(int M, int N, int K, int ksplit, int act, const float* W, int ldw, int nslab_in) {
    int nw, kw;
    return M >= 1 && M <= W4LN_MAX_ROWS && K == W4LN_D && N >= 16 && ksplit >= 1 && K % ksplit == 0 && w4_shape(K / ksplit, nw, kw) && kw == 64 &&
           ((((uintptr_t)W) & 15) == 0) && ldw % 4 == 0 && (ksplit == 1 || act == ACT_NONE) && (act == ACT_NONE || act == ACT_GELU || act == ACT_SILU) &&
           (nslab_in == 0 || nslab_in == 2 || nslab_in == 4);
}
// C = act(adaLN(res + bias_in + sum slab_in) W^T + bias) or raw K-slice slabs of it; see gemm_w4ln_kernel.  K = 512, M <= W4LN_MAX_ROWS.
int gemm_w4ln(const float* res, int ldres, const float* slab_in, int nslab_in, const float* bias_in, const float* mod, int ldmod, float* xout, int ldxo,
              const float* W, int ldw, const float* bias, int act, float* C, int ldc, float* slab, int M, int N, int K, int ksplit, hipStream_t s) {
    AID_REQUIRE(gemm_w4ln_ok(M, N, K, ksplit, act, W, ldw, nslab_in) && res && mod && (ksplit == 1 ? C != nullptr : slab != nullptr) && (nslab_in == 0 || slab_in) &&
                slab_in != slab, AID_E_SHAPE, "gemm_w4ln: unsupported M=%d N=%d K=%d ksplit=%d act=%d slabs=%d", M, N, K, ksplit, act, nslab_in);
    MidArgs g; memset(&g, 0, sizeof(g));
    g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.slab = slab; g.M = M; g.N = N; g.K = K; g.ksplit = ksplit; g.act = act;
    W4Ln ln{res, ldres, slab_in, (size_t)M * W4LN_D, bias_in, mod, ldmod, xout, ldxo};
    ProfScope prof(PROF_GEMM, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)ksplit * M * N));
    prof.m = M; prof.n = N; prof.k = K; prof.layout = act; prof.variant = 27;
    int nw = 0, kw = 0; w4_shape(K / ksplit, nw, kw);
    const dim3 grid(((N + 15) / 16) * ksplit);
    const size_t lds = sizeof(float) * ((size_t)32 * W4LN_LD + (size_t)nw * 2 * 64 * 4);
    auto go = [&](auto kern) {
        // 66 KB + the fold buffer: above the 64 KB a kernel gets without asking
        static thread_local const void* done[24]; static thread_local int ndone = 0;
        const void* fp = (const void*)kern;
        bool have = false;
        for (int i = 0; i < ndone; ++i) have = have || done[i] == fp;
        if (!have) { if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AID_E_HIP; if (ndone < 24) done[ndone++] = fp; }
        hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds, s, g, ln);
        return (int)AID_OK;
    };
    int rc;
#define AID_W4LN_S(NW_, ACT_, NS_) rc = go(gemm_w4ln_kernel<64, NW_, ACT_, NS_>)
#define AID_W4LN(NW_, ACT_) { if (nslab_in == 0) AID_W4LN_S(NW_, ACT_, 0); else if (nslab_in == 2) AID_W4LN_S(NW_, ACT_, 2); else AID_W4LN_S(NW_, ACT_, 4); }
    if (nw == 8) { if (act == ACT_GELU) AID_W4LN(8, ACT_GELU) else if (act == ACT_SILU) AID_W4LN(8, ACT_SILU) else AID_W4LN(8, ACT_NONE) }
    else { if (act == ACT_GELU) AID_W4LN(4, ACT_GELU) else if (act == ACT_SILU) AID_W4LN(4, ACT_SILU) else AID_W4LN(4, ACT_NONE) }
#undef AID_W4LN
#undef AID_W4LN_S
    if (rc != AID_OK) { ::aid::set_error("gemm_w4ln: hipFuncSetAttribute failed"); return rc; }
    AID_LAUNCH_CHECK();
    return AID_OK;
}

// ---- consumers of the split-K slabs ---------------------------------------------------------------------------------------------
// x_new = res + bias + sum_y slab[y]  (fixed order);  y = LN(x_new) (1 + scale) + shift with [scale | shift] = mod row.  One wave per
// row, D = 256 NV.  Writes x_new (the residual stream) and y.  Reference: DiTBlock.forward's `x = x + ...` followed by the next adaLN
// (models/score_networks.py:214-235, 257-270).
template <int NV, int NSLAB>
__global__ void adaln_sum_kernel(const float* __restrict__ res, int ldres, const float* __restrict__ slab, size_t slab_stride,
                                 const float* __restrict__ bias, const float* __restrict__ mod, int ldmod, float* __restrict__ xout, int ldxo,
                                 float* __restrict__ y, int ldy, int M) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int D = 256 * NV;
    AID_CHAIN_PRIO();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const f4* sc = reinterpret_cast<const f4*>(mod + (size_t)row * ldmod);
    const f4* sh = reinterpret_cast<const f4*>(mod + (size_t)row * ldmod + D);
    // every load of the row is issued before the first use: one memory round trip instead of NSLAB + 2 (768 waves on 256 CUs: nothing
    // else hides the latency)
    f4 v[NV], a[NV], b[NV], bi[NV], sl[NSLAB > 0 ? NSLAB : 1][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = reinterpret_cast<const f4*>(res + (size_t)row * ldres)[lane + 64 * i];
        a[i] = sc[lane + 64 * i]; b[i] = sh[lane + 64 * i];
        bi[i] = bias ? reinterpret_cast<const f4*>(bias)[lane + 64 * i] : (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int y0 = 0; y0 < NSLAB; ++y0)
#pragma unroll
        for (int i = 0; i < NV; ++i) sl[y0][i] = reinterpret_cast<const f4*>(slab + (size_t)y0 * slab_stride + (size_t)row * D)[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] += bi[i];
#pragma unroll
        for (int y0 = 0; y0 < NSLAB; ++y0) v[i] += sl[y0][i];          // fixed order: slab 0, 1, 2, ...
    }
    if (xout) {
#pragma unroll
        for (int i = 0; i < NV; ++i) reinterpret_cast<f4*>(xout + (size_t)row * ldxo)[lane + 64 * i] = v[i];
    }
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
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * (1.0f + a[i][j]) + b[i][j];
        reinterpret_cast<f4*>(y + (size_t)row * ldy)[lane + 64 * i] = o;
    }
}

// Block 0's adaLN with the latent projection folded in: x = z W_lp^T + b_lp (K = latent_dim <= 128: 17 FMAs per output at HalfCheetah dims, a
// launch of its own before), then y = LN(x) (1 + scale) + shift.  Wt is W_lp transposed to [L][D] once per reverse loop so that a lane's
// 4 consecutive columns are one 16-B load; the z row is a wave-uniform (broadcast) load.  Reference: models/score_networks.py:143 (latent_proj)
// + the first AdaptiveLayerNorm of block 0.
template <int NV>
__global__ void adaln_lp_kernel(const float* __restrict__ z, int L, const float* __restrict__ Wt, const float* __restrict__ blp,
                                const float* __restrict__ mod, int ldmod, float* __restrict__ xout, int ldxo, float* __restrict__ y, int ldy, int M) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    adaln_lp_row<NV>(z + (size_t)row * L, L, Wt, blp, mod + (size_t)row * ldmod, xout + (size_t)row * ldxo, y + (size_t)row * ldy, lane);
}
__global__ void transpose_small_kernel(const float* __restrict__ W /*[R][C]*/, float* __restrict__ Wt /*[C][R]*/, int R, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * C) { const int r = i / C, c = i % C; Wt[(size_t)c * R + r] = W[i]; }
}
int transpose_small(const float* W, float* Wt, int R, int C, hipStream_t s) {
    hipLaunchKernelGGL(transpose_small_kernel, dim3((R * C + 255) / 256), dim3(256), 0, s, W, Wt, R, C);
    AID_LAUNCH_CHECK(); return AID_OK;
}
int adaln_lp(const float* z, int L, const float* Wt, const float* blp, const float* mod, int ldmod, float* xout, int ldxo, float* y, int ldy, int M, int D,
             hipStream_t s) {
    AID_REQUIRE((D == 256 || D == 512 || D == 1024) && ((((uintptr_t)Wt | (uintptr_t)blp | (uintptr_t)mod | (uintptr_t)xout | (uintptr_t)y) & 15) == 0) &&
                ldmod % 4 == 0 && ldxo % 4 == 0 && ldy % 4 == 0, AID_E_SHAPE, "adaln_lp: D=%d or alignment unsupported", D);
    const dim3 grid((M + 3) / 4);
    if (D == 256) hipLaunchKernelGGL(adaln_lp_kernel<1>, grid, dim3(256), 0, s, z, L, Wt, blp, mod, ldmod, xout, ldxo, y, ldy, M);
    else if (D == 512) hipLaunchKernelGGL(adaln_lp_kernel<2>, grid, dim3(256), 0, s, z, L, Wt, blp, mod, ldmod, xout, ldxo, y, ldy, M);
    else hipLaunchKernelGGL(adaln_lp_kernel<4>, grid, dim3(256), 0, s, z, L, Wt, blp, mod, ldmod, xout, ldxo, y, ldy, M);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// The score head's last layer + the reverse step with the FIRST head layer taken from split-K slabs: p0 = SiLU(b0 + sum_y slab[y]) is formed
// in registers (fixed order), then out = p0 W2^T, clamp, scale, p_sample - one wave per row, as head2_p_sample's wave kernel (rowwise.hip).
// Reference: models/score_networks.py:164-171 + core/diffusion.py:208-255.
template <int KV, int NSLAB>
__global__ void __launch_bounds__(256) head2_slab_p_sample_kernel(const float* __restrict__ slab, size_t slab_stride, const float* __restrict__ b0,
                                                                  const float* __restrict__ W2, int Kp, const float* __restrict__ mult, float tv, int cont,
                                                                  const float* __restrict__ coef, int step, const float* __restrict__ z,
                                                                  const float* __restrict__ eps, float* __restrict__ zo, float* __restrict__ traj, int M, int L,
                                                                  NextLp nx) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float zs[4][128];                            // the new z rows of this block's four waves, for the optional next-step tail
    AID_CHAIN_PRIO();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 4 + wv;
    const bool live = row < M;
    if (live) {
        f4 pv[KV], sl[NSLAB][KV];
#pragma unroll
        for (int j = 0; j < KV; ++j) pv[j] = reinterpret_cast<const f4*>(b0)[lane + 64 * j];
#pragma unroll
        for (int y0 = 0; y0 < NSLAB; ++y0)
#pragma unroll
            for (int j = 0; j < KV; ++j) sl[y0][j] = reinterpret_cast<const f4*>(slab + (size_t)y0 * slab_stride + (size_t)row * Kp)[lane + 64 * j];
#pragma unroll
        for (int j = 0; j < KV; ++j) {
#pragma unroll
            for (int y0 = 0; y0 < NSLAB; ++y0) pv[j] += sl[y0][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[j][e] = act_apply(pv[j][e], ACT_SILU);
        }
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
                float scv = fminf(fmaxf(mine, -10.f), 10.f) * mult[0];
                if (cont) scv *= sqrtf(1.0f / (1e-5f + tv));
                const float zt = z[i];
                const float z0 = (zt + coef[0] * scv) * coef[1];
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
bool head2_slab_ok(int Kp, int nslab, const float* slab, const float* b0, const float* W2) {
    return (Kp == 256 || Kp == 512) && nslab == 4 && ((((uintptr_t)slab | (uintptr_t)b0 | (uintptr_t)W2) & 15) == 0);
}
int head2_slab_p_sample(const float* slab, int nslab, const float* b0, const float* W2, int Kp, const float* mult, float t_value, int continuous,
                        const float* coef, int step, const float* z, const float* eps, float* z_out, float* traj_out, int M, int L, hipStream_t s, const NextLp* next) {
    AID_REQUIRE(head2_slab_ok(Kp, nslab, slab, b0, W2) && L >= 1 && L <= 128, AID_E_SHAPE, "head2_slab_p_sample: Kp=%d nslab=%d L=%d", Kp, nslab, L);
    NextLp nx; memset(&nx, 0, sizeof(nx));
    if (next) { nx = *next; AID_REQUIRE(head2_next_ok(Kp, L, nx), AID_E_SHAPE, "head2_slab_p_sample: next-step tail unsupported (Kp=%d D=%d)", Kp, nx.D); }
    const size_t stride = (size_t)M * Kp;
    if (Kp == 256) hipLaunchKernelGGL((head2_slab_p_sample_kernel<1, 4>), dim3((M + 3) / 4), dim3(256), 0, s, slab, stride, b0, W2, Kp, mult, t_value, continuous,
                                      coef, step, z, eps, z_out, traj_out, M, L, nx);
    else hipLaunchKernelGGL((head2_slab_p_sample_kernel<2, 4>), dim3((M + 3) / 4), dim3(256), 0, s, slab, stride, b0, W2, Kp, mult, t_value, continuous,
                            coef, step, z, eps, z_out, traj_out, M, L, nx);
    AID_LAUNCH_CHECK(); return AID_OK;
}

bool adaln_sum_ok(int D, const float* res, int ldres, const float* slab, const float* bias, const float* mod, int ldmod, const float* xout, int ldxo,
                  const float* y, int ldy) {
    return (D == 256 || D == 512 || D == 1024) && ((((uintptr_t)res | (uintptr_t)slab | (uintptr_t)bias | (uintptr_t)mod | (uintptr_t)xout | (uintptr_t)y) & 15) == 0) &&
           ldres % 4 == 0 && ldmod % 4 == 0 && ldxo % 4 == 0 && ldy % 4 == 0;
}
template <int NV>
static int launch_adaln_sum(const float* res, int ldres, const float* slab, int nslab, const float* bias, const float* mod, int ldmod, float* xout, int ldxo,
                            float* y, int ldy, int M, hipStream_t s) {
    const dim3 grid((M + 3) / 4);
    const size_t stride = (size_t)M * 256 * NV;
    switch (nslab) {
        case 0: hipLaunchKernelGGL((adaln_sum_kernel<NV, 0>), grid, dim3(256), 0, s, res, ldres, slab, stride, bias, mod, ldmod, xout, ldxo, y, ldy, M); break;
        case 1: hipLaunchKernelGGL((adaln_sum_kernel<NV, 1>), grid, dim3(256), 0, s, res, ldres, slab, stride, bias, mod, ldmod, xout, ldxo, y, ldy, M); break;
        case 2: hipLaunchKernelGGL((adaln_sum_kernel<NV, 2>), grid, dim3(256), 0, s, res, ldres, slab, stride, bias, mod, ldmod, xout, ldxo, y, ldy, M); break;
        case 4: hipLaunchKernelGGL((adaln_sum_kernel<NV, 4>), grid, dim3(256), 0, s, res, ldres, slab, stride, bias, mod, ldmod, xout, ldxo, y, ldy, M); break;
        case 8: hipLaunchKernelGGL((adaln_sum_kernel<NV, 8>), grid, dim3(256), 0, s, res, ldres, slab, stride, bias, mod, ldmod, xout, ldxo, y, ldy, M); break;
        case 16: hipLaunchKernelGGL((adaln_sum_kernel<NV, 16>), grid, dim3(256), 0, s, res, ldres, slab, stride, bias, mod, ldmod, xout, ldxo, y, ldy, M); break;
        default: ::aid::set_error("adaln_sum: %d slabs (0, 1, 2, 4, 8 or 16 are built)", nslab); return AID_E_SHAPE;
    }
    AID_LAUNCH_CHECK();
    return AID_OK;
}
int adaln_sum(const float* res, int ldres, const float* slab, int nslab, const float* bias, const float* mod, int ldmod, float* xout, int ldxo,
              float* y, int ldy, int M, int D, hipStream_t s) {
    AID_REQUIRE(adaln_sum_ok(D, res, ldres, slab, bias, mod, ldmod, xout, ldxo, y, ldy), AID_E_SHAPE, "adaln_sum: D=%d or alignment unsupported", D);
    if (!slab) nslab = 0;
    if (D == 256) return launch_adaln_sum<1>(res, ldres, slab, nslab, bias, mod, ldmod, xout, ldxo, y, ldy, M, s);
    if (D == 512) return launch_adaln_sum<2>(res, ldres, slab, nslab, bias, mod, ldmod, xout, ldxo, y, ldy, M, s);
    return launch_adaln_sum<4>(res, ldres, slab, nslab, bias, mod, ldmod, xout, ldxo, y, ldy, M, s);
}

}  // namespace aid

using namespace aid;
extern "C" int aid_mid_debug(int bmt_min) {      // > 0: lowest tile height (units of 32 rows); < 0: highest (-1 .. -4); 0: defaults
    if (bmt_min < 0) g_mid_bmt_max = -bmt_min > 4 ? 4 : -bmt_min;
    else if (bmt_min == 0) { g_mid_bmt_min = 1; g_mid_bmt_max = 4; }
    else g_mid_bmt_min = bmt_min > 4 ? 4 : bmt_min;
    return AID_OK;
}
extern "C" {
int aid_gemm_mid(const float* A, int lda, const float* W, int ldw, const float* bias, int act, const float* residual, int ldr, float* C, int ldc,
                 float* slab, int M, int N, int K, int ksplit, void* stream) {
    AID_REQUIRE(A && W, AID_E_NULL, "aid_gemm_mid: null operand");
    return gemm_mid(A, lda, W, ldw, bias, act, residual, ldr, C, ldc, slab, M, N, K, ksplit, (hipStream_t)stream);
}
int aid_gemm_w4(const float* A, int lda, const float* W, int ldw, const float* bias, int act, float* C, int ldc, float* slab, int M, int N, int K, int ksplit,
                void* stream) {
    AID_REQUIRE(A && W, AID_E_NULL, "aid_gemm_w4: null operand");
    return gemm_w4(A, lda, W, ldw, bias, act, C, ldc, slab, M, N, K, ksplit, (hipStream_t)stream);
}
int aid_adaln_sum(const float* res, int ldres, const float* slab, int nslab, const float* bias, const float* mod, int ldmod, float* xout, int ldxo,
                  float* y, int ldy, int M, int D, void* stream) {
    AID_REQUIRE(res && mod && y, AID_E_NULL, "aid_adaln_sum: null tensor");
    return adaln_sum(res, ldres, slab, nslab, bias, mod, ldmod, xout, ldxo, y, ldy, M, D, (hipStream_t)stream);
}
}
