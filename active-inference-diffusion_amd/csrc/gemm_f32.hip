// fp32-in / fp32-accumulate MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), with fused epilogues.
//
//   C[M,N] = act(alpha * sum_k A(m,k) * B(n,k) + bias[n]) * mask + residual   (+ C if accumulate)
//
// This one kernel family replaces every dense contraction of the hot path: nn.Linear forward
// (A=[M][K] activations, B=[N][K] weight), its data gradient (B read k-major) and its weight
// gradient (both operands read k-major); reference call sites: aid/models/score_networks.py:41-99,
// policy_networks.py:42-71, value_networks.py:24-45, dynamics_models.py:28-45,
// core/active_inference.py:111-171,886-919.
//
// Design (MI355X): exact-f32 MFMA runs at the f32 vector peak (157 TFLOP/s) and needs ONE f32 per
// lane per operand per instruction, so LDS traffic is light; what matters is (a) keeping global
// loads wide and coalesced along whichever dimension is contiguous in memory and (b) exposing
// enough workgroups for batch-256 problems.  Each operand has two LDS images chosen by its memory
// layout: k-contiguous operands are staged as [row][BK+4] (b128 writes, b128 fragment reads,
// conflict-free by the +4 pad) and k-major operands as [k][rows+4] (b128 writes, b32 fragment
// reads across consecutive lanes).  Because a b128 fragment read hands a lane four consecutive k,
// the k order inside an 8-deep group is permuted identically for both operands (the reduction is
// order-free): MFMA j of group g contracts k = {8g+j, 8g+4+j}.
//
// Tiles: 128x128 (4 waves as 2x2, 64x64 per wave) for large M, 64x64 (32x32 per wave) and
// 32x32 + 4-way in-workgroup split-K for the batch<=512 layers where the grid would otherwise
// leave most of the 256 CUs idle.  Register-staged double buffering: tile t+1's global loads are
// in flight while tile t is contracted.
#include <atomic>
#include <mutex>
#include "aid_common.h"
#include "rowln.h"
#include "rowwise.h"

namespace aid {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

template <int ROWS, bool KMAJOR>
struct Tile {  // LDS image of one operand tile (ROWS = BM or BN)
    static constexpr int LD = KMAJOR ? (ROWS + 4) : (BK + 4);
    static constexpr int NROW = KMAJOR ? BK : ROWS;
    static constexpr int FLOATS = LD * NROW;
    static constexpr int F4 = ROWS * BK / 4;  // float4 per tile
};

// global -> registers for one operand tile. P = float4 per thread.
template <int ROWS, bool KMAJOR, bool VEC, int NT, int P>
__device__ __forceinline__ void load_tile(const float* __restrict__ G, int ld, int r0, int k0, int R, int K,
                                          int tid, f32x4 (&regs)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int f = tid + NT * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f < Tile<ROWS, KMAJOR>::F4) {
            if (!KMAJOR) {
                const int row = f / (BK / 4), kq = f % (BK / 4);
                const int gr = r0 + row, gk = k0 + 4 * kq;
                if (gr < R) {
                    const float* p = G + (size_t)gr * ld + gk;
                    if (VEC) { if (gk < K) v = *reinterpret_cast<const f32x4*>(p); }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gk + j < K) v[j] = p[j];
                    }
                }
            } else {
                const int k = f / (ROWS / 4), rq = f % (ROWS / 4);
                const int gk = k0 + k, gr = r0 + 4 * rq;
                if (gk < K) {
                    const float* p = G + (size_t)gk * ld + gr;
                    if (VEC) { if (gr < R) v = *reinterpret_cast<const f32x4*>(p); }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gr + j < R) v[j] = p[j];
                    }
                }
            }
        }
        regs[i] = v;
    }
}

template <int ROWS, bool KMAJOR, int NT, int P>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const f32x4 (&regs)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int f = tid + NT * i;
        if (f < Tile<ROWS, KMAJOR>::F4) {
            int off;
            if (!KMAJOR) { const int row = f / (BK / 4), kq = f % (BK / 4); off = row * Tile<ROWS, KMAJOR>::LD + 4 * kq; }
            else { const int k = f / (ROWS / 4), rq = f % (ROWS / 4); off = k * Tile<ROWS, KMAJOR>::LD + 4 * rq; }
            *reinterpret_cast<f32x4*>(S + off) = regs[i];
        }
    }
}

// fragment of 4 k-values (group g8, this lane's half h) for row `r` of the tile
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ S, int r, int g8, int h) {
    if (!KMAJOR) return *reinterpret_cast<const f32x4*>(S + r * Tile<ROWS, KMAJOR>::LD + 8 * g8 + 4 * h);
    f32x4 v;
    const float* p = S + (8 * g8 + 4 * h) * Tile<ROWS, KMAJOR>::LD + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = p[j * Tile<ROWS, KMAJOR>::LD];
    return v;
}

// blockIdx -> output tile.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the XCD group: speed only, never
// correctness).  When the N-tile count is a multiple of 8, all M-tiles of one weight panel (same tn) are given consecutive
// slots of ONE XCD group, so the panel is fetched from the Infinity Cache once per GEMM and the other M-tiles hit that XCD's L2
// (batch-256 layers stream ~55 MB of weights per denoise step through 4 MB L2s).  Bijective for any grid.
__device__ __forceinline__ void tile_coords(int b, int tiles_m, int tiles_n, int& tm, int& tn) {
    if ((tiles_n & 7) == 0) { const int x = b & 7, q = b >> 3; tn = x + 8 * (q / tiles_m); tm = q % tiles_m; }
    else { tm = b / tiles_n; tn = b % tiles_n; }
}

__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int row, int col, float v) {
    if (row >= g.M || col >= g.N) return;
    v *= g.alpha;
    if (g.bias) v += g.bias[col];
    const size_t o = (size_t)row * g.ldc + col;
    if (g.preact) g.preact[o] = v;
    v = act_apply(v, g.act);
    if (g.dact_pre) v *= act_grad(g.dact_pre[o], g.dact);
    if (g.mask) v *= g.mask[o];
    if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
    if (g.accumulate) v += g.C[o];
    g.C[o] = v;
}

// ------------------------------------------------------------------------------------------------
// Main kernel: BMxBN tile, 4 waves as 2x2, each wave (BM/2)x(BN/2) = TMxTN MFMA tiles of 32x32.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool AKM, bool BKM, bool VEC>
__global__ void __launch_bounds__(256) gemm_tile_kernel(GemmArgs g, int tiles_n) {
    constexpr int NT = 256;
    constexpr int TM = BM / 64, TN = BN / 64;
    using TA = Tile<BM, AKM>; using TB = Tile<BN, BKM>;
    constexpr int PA = (TA::F4 + NT - 1) / NT, PB = (TB::F4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // 2 stages
    float* Bs = smem + 2 * TA::FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    int tm_, tn_; tile_coords(blockIdx.x, (g.M + BM - 1) / BM, tiles_n, tm_, tn_);
    const int m0 = tm_ * BM, n0 = tn_ * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[PA], rb[PB];
    const int nk_all = (g.K + BK - 1) / BK;
    const int per = (nk_all + gridDim.y - 1) / gridDim.y;           // K tiles per slice (gridDim.y = 1 unless slab mode)
    const int kt0 = blockIdx.y * per;
    const int nk = max(0, min(per, nk_all - kt0));
    load_tile<BM, AKM, VEC, NT, PA>(g.A, g.lda, m0, kt0 * BK, g.M, g.K, tid, ra);
    load_tile<BN, BKM, VEC, NT, PB>(g.B, g.ldb, n0, kt0 * BK, g.N, g.K, tid, rb);
    store_tile<BM, AKM, NT, PA>(As, tid, ra);
    store_tile<BN, BKM, NT, PB>(Bs, tid, rb);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<BM, AKM, VEC, NT, PA>(g.A, g.lda, m0, (kt0 + kt + 1) * BK, g.M, g.K, tid, ra);
            load_tile<BN, BKM, VEC, NT, PB>(g.B, g.ldb, n0, (kt0 + kt + 1) * BK, g.N, g.K, tid, rb);
        }
        const float* a_s = As + cur * TA::FLOATS;
        const float* b_s = Bs + cur * TB::FLOATS;
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
            f32x4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = read_frag<BM, AKM>(a_s, wm + 32 * i + li, g8, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = read_frag<BN, BKM>(b_s, wn + 32 * j + li, g8, h);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile<BM, AKM, NT, PA>(As + (cur ^ 1) * TA::FLOATS, tid, ra);
            store_tile<BN, BKM, NT, PB>(Bs + (cur ^ 1) * TB::FLOATS, tid, rb);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // (full unroll is mandatory: a rolled loop would index acc[][] at run time and send the accumulators to scratch)
#pragma clang loop unroll(full)
    for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
        for (int j = 0; j < TN; ++j)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = n0 + wn + 32 * j + li;
                if (g.slab) { if (row < g.M && col < g.N) g.slab[((size_t)blockIdx.y * g.M + row) * g.N + col] = acc[i][j][r]; }
                else epilogue_store(g, row, col, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------
// K <= 32 (the first layer of every network on the path: K = latent_dim / observation_dim = 17 at HalfCheetah sizes).
// An MFMA tile kernel pads such a K to a 32-deep stage and pays its whole staging pipeline for a handful of FMAs; here
// a 64x64 output tile per workgroup keeps both operand tiles (64 x K each) in LDS and every thread owns 4x4 outputs on
// the f32 VALU.  The kernel is output-write bound (M*N*4 bytes), which is the right bound for this shape.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gemm_smallk_kernel(GemmArgs g, int tiles_n) {
    __shared__ float xs[64][33];
    __shared__ float wsm[64][33];
    const int tid = threadIdx.x;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 64, n0 = tn * 64, K = g.K;
    for (int i = tid; i < 64 * K; i += 256) {
        const int r = i / K, k = i % K;
        xs[r][k] = (m0 + r < g.M) ? g.A[(size_t)(m0 + r) * g.lda + k] : 0.f;
        if (!g.b_kmajor) wsm[r][k] = (n0 + r < g.N) ? g.B[(size_t)(n0 + r) * g.ldb + k] : 0.f;
    }
    if (g.b_kmajor)      // W stored [K][N] (the data gradient through a layer with <= 32 outputs): n runs fastest in memory
        for (int i = tid; i < 64 * K; i += 256) { const int k = i >> 6, r = i & 63; wsm[r][k] = (n0 + r < g.N) ? g.B[(size_t)k * g.ldb + n0 + r] : 0.f; }
    __syncthreads();
    const int tc = tid & 15, tr = tid >> 4;      // rows tr + 16 i, columns 4 tc + j (contiguous: 16-B stores on the plain epilogue)
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k = 0; k < K; ++k) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = xs[tr + 16 * i][k];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = wsm[4 * tc + j][k];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    const int c0 = n0 + 4 * tc;
    const bool plain = !g.preact && !g.dact_pre && !g.mask && !g.residual && !g.accumulate && (g.ldc & 3) == 0 && c0 + 3 < g.N &&
                       ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
    if (plain) {
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = g.bias[c0 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + tr + 16 * i;
            if (row >= g.M) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_apply(acc[i][j] * g.alpha + bb[j], g.act);
            *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + c0) = o;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) epilogue_store(g, m0 + tr + 16 * i, c0 + j, acc[i][j]);
}
static thread_local int t_variant = 0;    // code of the kernel the last launch_* call used (profiling label: aid_prof_dump's `variant` column)
static int launch_smallk(const GemmArgs& g, hipStream_t s) {
    t_variant = 3;
    const int tm = (g.M + 63) / 64, tn = (g.N + 63) / 64;
    hipLaunchKernelGGL(gemm_smallk_kernel, dim3(tm * tn), dim3(256), 0, s, g, tn);
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------
// Large-problem variant of the tile kernel with the K depth of a stage (BKT) and the workgroup->tile map as
// parameters.  Shallower stages (BKT = 16) shrink the LDS image so that 128-row tiles still run 3-4 workgroups
// per CU, and a 128-wide tile halves the bytes each CU must ingest per MFMA (the per-CU L2/MALL ingest rate, not
// HBM, is what starves fp32 MFMA at 64x64).  map = 1: each XCD works through all N-tiles of a group of 8 for one
// M-tile before moving to the next M-tile, so an activation tile is fetched from the Infinity Cache once per XCD
// and re-used from that XCD's L2, while the 8 weight panels of the group (<= 2 MB at K = 1024) stay L2-resident.
// ------------------------------------------------------------------------------------------------
template <int ROWS, bool KMAJOR, int BKT>
struct TileK {
    static constexpr int LD = KMAJOR ? (ROWS + 4) : (BKT + 4);
    static constexpr int NROW = KMAJOR ? BKT : ROWS;
    static constexpr int FLOATS = LD * NROW;
    static constexpr int F4 = ROWS * BKT / 4;
};

// Branch-free: every lane always loads from an in-range (clamped) address; out-of-range values are zeroed when the tile is written
// to LDS (store_tile_k), i.e. only once the data is needed.  A branch around a load - or a select on the loaded value next to the
// load - makes hipcc wait for the whole vector-memory queue (`s_waitcnt vmcnt(0)`) far too early, which would serialise the
// two-tile-deep prefetch below (cdna_hip_programming.md, "Projection GEMM at M = 256", trap 4c).
template <int ROWS, bool KMAJOR, int BKT, int NT, int P>
__device__ __forceinline__ void load_tile_k(const float* __restrict__ G, int ld, int r0, int k0, int R, int K, int tid, f32x4 (&regs)[P]) {
    static_assert(TileK<ROWS, KMAJOR, BKT>::F4 % NT == 0, "tile must be a whole number of float4 per thread");
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int f = tid + NT * i;
        const float* p;
        if (!KMAJOR) {
            const int row = f / (BKT / 4), kq = f % (BKT / 4);
            const int gr = r0 + row, gk = k0 + 4 * kq;
            p = G + (size_t)(gr < R ? gr : R - 1) * ld + (gk < K ? gk : 0);
        } else {
            const int k = f / (ROWS / 4), rq = f % (ROWS / 4);
            const int gk = k0 + k, gr = r0 + 4 * rq;
            p = G + (size_t)(gk < K ? gk : K - 1) * ld + (gr < R ? gr : 0);
        }
        regs[i] = *reinterpret_cast<const f32x4*>(p);
    }
}
template <int ROWS, bool KMAJOR, int BKT, int NT, int P>
__device__ __forceinline__ void store_tile_k(float* __restrict__ S, int r0, int k0, int R, int K, int tid, const f32x4 (&regs)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int f = tid + NT * i;
        int off; bool ok;
        if (!KMAJOR) { const int row = f / (BKT / 4), kq = f % (BKT / 4); off = row * TileK<ROWS, KMAJOR, BKT>::LD + 4 * kq;
                       ok = (r0 + row < R) && (k0 + 4 * kq < K); }
        else { const int k = f / (ROWS / 4), rq = f % (ROWS / 4); off = k * TileK<ROWS, KMAJOR, BKT>::LD + 4 * rq;
               ok = (k0 + k < K) && (r0 + 4 * rq < R); }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(S + off) = ok ? regs[i] : z;
    }
}
template <int ROWS, bool KMAJOR, int BKT>
__device__ __forceinline__ f32x4 read_frag_k(const float* __restrict__ S, int r, int g8, int h) {
    if (!KMAJOR) return *reinterpret_cast<const f32x4*>(S + r * TileK<ROWS, KMAJOR, BKT>::LD + 8 * g8 + 4 * h);
    f32x4 v;
    const float* p = S + (8 * g8 + 4 * h) * TileK<ROWS, KMAJOR, BKT>::LD + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = p[j * TileK<ROWS, KMAJOR, BKT>::LD];
    return v;
}

__device__ __forceinline__ void tile_coords_big(int b, int tiles_m, int tiles_n, int map, int& tm, int& tn) {
    if (map == 1) {
        const int G = tiles_n < 8 ? tiles_n : 8;
        if (tiles_n % G == 0) {
            // task = (M-tile, group of G column tiles); whole rounds of 8 tasks are dealt one per XCD (b % 8), so the G workgroups that
            // share an activation row tile run back to back on ONE XCD; the < 8 tasks left over are laid out plainly.  Bijective for any
            // grid.  (Without it a 29 952 x 13 312 x 512 launch pulled 13.6 GB through the L2s for 88 MB of operands: every row of
            // tiles re-read the whole 27 MB weight matrix.)
            const int ntasks = tiles_m * (tiles_n / G), n8 = ntasks & ~7, full = n8 * G;
            int T, j;
            if (b < full) { const int x = b & 7, q = b >> 3; T = x + 8 * (q / G); j = q % G; }
            else { const int r = b - full; T = n8 + r / G; j = r % G; }
            tm = T % tiles_m; tn = (T / tiles_m) * G + j;
            return;
        }
    }
    tile_coords(b, tiles_m, tiles_n, tm, tn);
}

// FL > 0 (slab mode, PD == 2 only): "deep-K" form for contractions hundreds of thousands deep (the encoder's Linear(451 584 -> 2L)).  The MFMA
// accumulators are added into a second accumulator set and cleared every FL stages (FL x BKT k-values), so no float32 sum in the kernel is a chain of
// more than FL x BKT roundings followed by a chain of (slice depth / (FL x BKT)) - a blocked summation whose error grows with the square root of those
// lengths instead of the slice depth - and the slices are folded in float64 (slab_fold_kernel).  The grid is 1-D with the workgroups of one K slice
// dealt to ONE XCD (b % 8), so the A / B slice a 128 x 128 tile shares with its neighbours is fetched through that XCD's L2 once.
template <int BM, int BN, int BKT, bool AKM, bool BKM, int PD, int FL = 0>
__global__ void __launch_bounds__(256) gemm_big_kernel(GemmArgs g, int tiles_n, int map) {
    constexpr int NT = 256;
    static_assert(FL == 0 || (PD == 2 && FL % 2 == 0), "the flush sits behind the odd step of the two-stage loop");
    constexpr int TM = BM / 64, TN = BN / 64;
    using TA = TileK<BM, AKM, BKT>; using TB = TileK<BN, BKM, BKT>;
    constexpr int PA = (TA::F4 + NT - 1) / NT, PB = (TB::F4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * TA::FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    int tm_, tn_, slice = blockIdx.y, nslice = gridDim.y;
    if constexpr (FL > 0) {
        const int tiles = ((g.M + BM - 1) / BM) * tiles_n, b = blockIdx.x;
        int t;
        nslice = g.nslab;
        if (nslice % 8 == 0) { const int i = b >> 3; t = i % tiles; slice = (i / tiles) * 8 + (b & 7); }
        else { t = b % tiles; slice = b / tiles; }
        tm_ = t / tiles_n; tn_ = t % tiles_n;
    } else
        tile_coords_big(blockIdx.x, (g.M + BM - 1) / BM, tiles_n, map, tm_, tn_);
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    f32x16 acc[TM][TN];
    f32x16 acc2[FL > 0 ? TM : 1][FL > 0 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if constexpr (FL > 0) acc2[i][j][r] = 0.f; }
    // slab mode (deterministic split-K): a slice owns a contiguous range of K tiles and writes its partial tile to slab[slice]
    const int nk_all = (g.K + BKT - 1) / BKT;
    const int per = (nk_all + nslice - 1) / nslice;
    const int ktb = slice * per;                                       // first K tile of this slice
    const int nk = max(0, min(per, nk_all - ktb));
    const int kend = min(g.K, (ktb + nk) * BKT);                       // loads beyond this slice's K range come back as zeros
    auto contract = [&](int cur) {
        const float* a_s = As + cur * TA::FLOATS;
        const float* b_s = Bs + cur * TB::FLOATS;
#pragma unroll
        for (int g8 = 0; g8 < BKT / 8; ++g8) {
            f32x4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = read_frag_k<BM, AKM, BKT>(a_s, wm + 32 * i + li, g8, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = read_frag_k<BN, BKM, BKT>(b_s, wn + 32 * j + li, g8, h);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (PD == 1) {
        f32x4 ra[PA], rb[PB];
        load_tile_k<BM, AKM, BKT, NT, PA>(g.A, g.lda, m0, ktb * BKT, g.M, kend, tid, ra);
        load_tile_k<BN, BKM, BKT, NT, PB>(g.B, g.ldb, n0, ktb * BKT, g.N, kend, tid, rb);
        store_tile_k<BM, AKM, BKT, NT, PA>(As, m0, ktb * BKT, g.M, kend, tid, ra);
        store_tile_k<BN, BKM, BKT, NT, PB>(Bs, n0, ktb * BKT, g.N, kend, tid, rb);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                load_tile_k<BM, AKM, BKT, NT, PA>(g.A, g.lda, m0, (ktb + kt + 1) * BKT, g.M, kend, tid, ra);
                load_tile_k<BN, BKM, BKT, NT, PB>(g.B, g.ldb, n0, (ktb + kt + 1) * BKT, g.N, kend, tid, rb);
            }
            contract(cur);
            if (kt + 1 < nk) {
                store_tile_k<BM, AKM, BKT, NT, PA>(As + (cur ^ 1) * TA::FLOATS, m0, (ktb + kt + 1) * BKT, g.M, kend, tid, ra);
                store_tile_k<BN, BKM, BKT, NT, PB>(Bs + (cur ^ 1) * TB::FLOATS, n0, (ktb + kt + 1) * BKT, g.N, kend, tid, rb);
            }
            __syncthreads();
        }
    } else {
        // prefetch distance 2: tile kt+2's global loads are issued before tile kt is contracted and land in the register set
        // that tile kt+1 does not use, so a load has two contraction periods to arrive (Infinity-Cache / HBM latency is longer
        // than one 32-deep stage of fp32 MFMAs).  The loop is unrolled by two so both register sets are named statically.
        f32x4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
        load_tile_k<BM, AKM, BKT, NT, PA>(g.A, g.lda, m0, ktb * BKT, g.M, kend, tid, ra0);
        load_tile_k<BN, BKM, BKT, NT, PB>(g.B, g.ldb, n0, ktb * BKT, g.N, kend, tid, rb0);
        load_tile_k<BM, AKM, BKT, NT, PA>(g.A, g.lda, m0, (ktb + 1) * BKT, g.M, kend, tid, ra1);     // rows beyond K come back as zeros
        load_tile_k<BN, BKM, BKT, NT, PB>(g.B, g.ldb, n0, (ktb + 1) * BKT, g.N, kend, tid, rb1);
        store_tile_k<BM, AKM, BKT, NT, PA>(As, m0, ktb * BKT, g.M, kend, tid, ra0);
        store_tile_k<BN, BKM, BKT, NT, PB>(Bs, n0, ktb * BKT, g.N, kend, tid, rb0);
        __syncthreads();
        // An odd tile count runs one extra all-zero tile (loads beyond K come back as zeros) instead of leaving the loop half way:
        // an exit between the two halves makes the compiler shuttle the accumulators between two register banks every iteration.
        const int nk2 = (nk + 1) & ~1;
        for (int kt = 0; kt < nk2; kt += 2) {
            // even step: LDS[0] holds tile kt, set 1 holds tile kt+1 (in flight), set 0 is free
            load_tile_k<BM, AKM, BKT, NT, PA>(g.A, g.lda, m0, (ktb + kt + 2) * BKT, g.M, kend, tid, ra0);
            load_tile_k<BN, BKM, BKT, NT, PB>(g.B, g.ldb, n0, (ktb + kt + 2) * BKT, g.N, kend, tid, rb0);
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of the MFMA block (the scheduler otherwise sinks it to the store)
            contract(0);
            store_tile_k<BM, AKM, BKT, NT, PA>(As + TA::FLOATS, m0, (ktb + kt + 1) * BKT, g.M, kend, tid, ra1);
            store_tile_k<BN, BKM, BKT, NT, PB>(Bs + TB::FLOATS, n0, (ktb + kt + 1) * BKT, g.N, kend, tid, rb1);
            __syncthreads();
            // odd step: LDS[1] holds tile kt+1, set 0 holds tile kt+2 (in flight), set 1 is free
            load_tile_k<BM, AKM, BKT, NT, PA>(g.A, g.lda, m0, (ktb + kt + 3) * BKT, g.M, kend, tid, ra1);
            load_tile_k<BN, BKM, BKT, NT, PB>(g.B, g.ldb, n0, (ktb + kt + 3) * BKT, g.N, kend, tid, rb1);
            __builtin_amdgcn_sched_barrier(0);
            contract(1);
            store_tile_k<BM, AKM, BKT, NT, PA>(As, m0, (ktb + kt + 2) * BKT, g.M, kend, tid, ra0);
            store_tile_k<BN, BKM, BKT, NT, PB>(Bs, n0, (ktb + kt + 2) * BKT, g.N, kend, tid, rb0);
            if constexpr (FL > 0) {
                if ((kt + 2) % FL == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { acc2[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
                }
            }
            __syncthreads();
        }
    }
    if constexpr (FL > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
        for (int j = 0; j < TN; ++j)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r)
            {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wn + 32 * j + li;
                if (g.slab) { if (row < g.M && col < g.N) g.slab[((size_t)slice * g.M + row) * g.N + col] = acc[i][j][r]; }
                else epilogue_store(g, row, col, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------
// Small-M kernel: 32x32 output tile per workgroup, the 4 waves split every 32-deep K tile four
// ways (8 k each = 4 MFMAs) and reduce through LDS.  4x more workgroups than the 64x64 tiling and
// a 4x shorter dependent MFMA chain per wave: this is the batch-256 configuration.
// ------------------------------------------------------------------------------------------------
template <bool AKM, bool BKM, bool VEC>
__global__ void __launch_bounds__(256) gemm_splitk_kernel(GemmArgs g, int tiles_n) {
    constexpr int NT = 256, BM = 32, BN = 32;
    using TA = Tile<BM, AKM>; using TB = Tile<BN, BKM>;
    constexpr int PA = (TA::F4 + NT - 1) / NT, PB = (TB::F4 + NT - 1) / NT;  // = 1
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * TA::FLOATS;
    float* red = smem + 2 * TA::FLOATS + 2 * TB::FLOATS;  // [3][16][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    int tm_, tn_; tile_coords(blockIdx.x, (g.M + BM - 1) / BM, tiles_n, tm_, tn_);
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[PA], rb[PB];
    const int nk = (g.K + BK - 1) / BK;
    load_tile<BM, AKM, VEC, NT, PA>(g.A, g.lda, m0, 0, g.M, g.K, tid, ra);
    load_tile<BN, BKM, VEC, NT, PB>(g.B, g.ldb, n0, 0, g.N, g.K, tid, rb);
    store_tile<BM, AKM, NT, PA>(As, tid, ra);
    store_tile<BN, BKM, NT, PB>(Bs, tid, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<BM, AKM, VEC, NT, PA>(g.A, g.lda, m0, (kt + 1) * BK, g.M, g.K, tid, ra);
            load_tile<BN, BKM, VEC, NT, PB>(g.B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, tid, rb);
        }
        const f32x4 av = read_frag<BM, AKM>(As + cur * TA::FLOATS, li, wave, h);
        const f32x4 bv = read_frag<BN, BKM>(Bs + cur * TB::FLOATS, li, wave, h);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        if (kt + 1 < nk) {
            store_tile<BM, AKM, NT, PA>(As + (cur ^ 1) * TA::FLOATS, tid, ra);
            store_tile<BN, BKM, NT, PB>(Bs + (cur ^ 1) * TB::FLOATS, tid, rb);
        }
        __syncthreads();
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane] + red[(2 * 16 + r) * 64 + lane];
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            epilogue_store(g, row, n0 + li, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Skinny kernel (act() / collector batches, M <= 32 rows): y = x W^T is a weight-streaming problem — N*K*4 bytes of W against a few KB of
// x — so every wave owns ONE output column: its W row sits in registers (KV float4 per lane, K <= 256 KV), the x rows come through L1 in
// 8-row chunks, each lane keeps 8 partial dots and the wave folds them with a fixed xor butterfly.  N waves fill the chip where the MFMA
// panel kernel has N/32 workgroups walking K serially (K = 2048: 14 us -> 3.5 us at M = 1).  VALU fp32 FMA, fixed summation order.
// ------------------------------------------------------------------------------------------------
template <int KV>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= g.N) return;                       // no barrier below
    f32x4 w[KV];
    int koff[KV];
#pragma unroll
    for (int j = 0; j < KV; ++j) {
        const int k = (j * 64 + lane) * 4;
        koff[j] = min(k, g.K - 4);              // clamped address, value zeroed below: the loads stay unconditional
        w[j] = *reinterpret_cast<const f32x4*>(g.B + (size_t)n * g.ldb + koff[j]);
        if (k >= g.K) w[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    constexpr int RB = KV >= 16 ? 1 : (16 / KV > 8 ? 8 : 16 / KV);      // rows whose x fragments are loaded together (<= 16 float4 in flight)
    for (int m0 = 0; m0 < g.M; m0 += 8) {
        float acc[8];
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += RB) {
            f32x4 x[RB][KV];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float* xr = g.A + (size_t)min(m0 + r0 + r, g.M - 1) * g.lda;
#pragma unroll
                for (int j = 0; j < KV; ++j) x[r][j] = *reinterpret_cast<const f32x4*>(xr + koff[j]);
            }
            __builtin_amdgcn_sched_barrier(0);           // all loads of the batch issued before the first FMA waits
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < KV; ++j) {
                    a = fmaf(x[r][j][0], w[j][0], a); a = fmaf(x[r][j][1], w[j][1], a); a = fmaf(x[r][j][2], w[j][2], a); a = fmaf(x[r][j][3], w[j][3], a);
                }
                acc[r0 + r] = a;
            }
        }
        // transpose-reduce: each butterfly stage halves the values a lane carries (4 + 2 + 1 + 3 = 10 cross-lane moves for 8 row sums);
        // afterwards lane 8 c (c = 0..7) holds the sum of row  4*(c>>2) + 2*((c>>1)&1) + (c&1)  =  c
        float a4[4], a2[2], a1;
        const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float recv = __shfl_xor(h32 ? acc[i] : acc[i + 4], 32, 64); a4[i] = (h32 ? acc[i + 4] : acc[i]) + recv; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { const float recv = __shfl_xor(h16 ? a4[i] : a4[i + 2], 16, 64); a2[i] = (h16 ? a4[i + 2] : a4[i]) + recv; }
        { const float recv = __shfl_xor(h8 ? a2[0] : a2[1], 8, 64); a1 = (h8 ? a2[1] : a2[0]) + recv; }
        a1 += __shfl_xor(a1, 4, 64); a1 += __shfl_xor(a1, 2, 64); a1 += __shfl_xor(a1, 1, 64);
        const int row = m0 + (lane >> 3);
        if ((lane & 7) == 0 && row < g.M) epilogue_store(g, row, n, a1);
    }
}
template <int KV>
static int launch_skinny_kv(const GemmArgs& g, hipStream_t s) {
    hipLaunchKernelGGL((gemm_skinny_kernel<KV>), dim3((g.N + 3) / 4), dim3(256), 0, s, g);
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------
// Thin weight-gradient kernel: C[M,N] (+)= A^T B with both operands k-major ([K][M], [K][N]), a handful of 32x32 output tiles and a long K
// (dW of a 6- or 17-wide layer over 2 560 rows).  One 1024-thread workgroup per tile: each of the 16 waves contracts its own K/16 slice
// straight from global memory (k-major operands are lane-contiguous along m / n, so a fragment is one coalesced dword load) and the
// slices are folded through LDS in wave order — a 16x shorter serial chain than one wave walking all of K, no workspace, reproducible.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) gemm_thin_tn_kernel(GemmArgs g, int tiles_n) {
    constexpr int NW = 16;
    __shared__ float red[(NW - 1) * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    int kper = (g.K + NW - 1) / NW; kper += kper & 1;
    const int kb = wave * kper, ke = min(g.K, kb + kper);
    const bool mok = m0 + li < g.M, nok = n0 + li < g.N;
    const float* pa = g.A + (mok ? m0 + li : 0);
    const float* pb = g.B + (nok ? n0 + li : 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = kb; k0 < ke; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = min(k0 + 2 * j + h, g.K - 1);
            av[j] = pa[(size_t)k * g.lda]; bv[j] = pb[(size_t)k * g.ldb];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = k0 + 2 * j + h < ke;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok && mok ? av[j] : 0.f, ok && nok ? bv[j] : 0.f, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            for (int w = 0; w < NW - 1; ++w) v += red[(w * 16 + r) * 64 + lane];
            epilogue_store(g, m0 + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + li, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Panel kernel (the batch<=512 workhorse): 32x32 output tile, K streamed in 256-deep stages through a
// 2-slot LDS ring by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, every byte fetched once per
// workgroup in whole 128-B lines).  All of a stage's DMA is issued at once, so a K=512 layer has its
// entire working set in flight before the first MFMA; deeper K keeps one stage (32 MFMAs per wave)
// ahead.  The 4 waves split each stage's K four ways (one wave per SIMD) and reduce through LDS.
// Synchronisation follows the CDNA4 rules for LDS-DMA: counted s_waitcnt vmcnt(N) by the issuing wave,
// then a raw s_barrier before any wave reads the slot; a slot is re-filled only after a barrier that
// every reader has passed.  One workgroup per CU (133 KB LDS) - these grids have <= ~4 tiles per CU.
// Requirements (checked by the dispatcher): K % 256 == 0, 16-B aligned rows; k-major operands need
// rows % 32 == 0.  k-contiguous operands clamp out-of-range rows (results discarded in the epilogue).
// ------------------------------------------------------------------------------------------------
constexpr int KS = 256;                      // k per stage
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;
#ifndef AID_PANEL_B_AUX
#define AID_PANEL_B_AUX 2      // cache-policy bits of the B-operand (weights) LDS-DMA loads of the panel / ring kernels: 2 = nt.  Measured
                               // against 0 (default policy) in one gpurun call: reverse loop at B = 256 24.8 -> 24.2 ms, train_step unchanged
#endif

template <bool KMAJOR>
struct Panel {  // LDS image of one 32-row operand stage
    static constexpr int LD = KMAJOR ? 32 : (KS + 4);
    static constexpr int FLOATS = KMAJOR ? KS * 32 : 32 * (KS + 4);
    static constexpr int DMA_PER_WAVE = 8;  // wave-instructions per stage per wave (32 total)
};

// issue this wave's share of one operand stage: 8 x global_load_lds_dwordx4 (1 KiB each)
template <bool KMAJOR, int AUX = 0>
__device__ __forceinline__ void panel_dma(const float* __restrict__ G, int ld, int r0, int R, int k0, float* slot,
                                          int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* src; float* dst;
        if (!KMAJOR) {  // one row (256 floats = 1 KiB) per instruction
            const int r = wave * 8 + j;
            int gr = r0 + r; gr = gr < R ? gr : R - 1;
            src = G + (size_t)gr * ld + k0 + 4 * lane;
            dst = slot + r * Panel<false>::LD;
        } else {        // 8 k-rows x 128 B per instruction
            const int kb = (wave * 8 + j) * 8;
            src = G + (size_t)(k0 + kb + (lane >> 3)) * ld + r0 + 4 * (lane & 7);
            dst = slot + kb * 32;
        }
        __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)dst, 16, 0, AUX);
    }
}

template <bool KMAJOR>
__device__ __forceinline__ f32x4 panel_frag(const float* __restrict__ slot, int li, int k, int h) {
    if (!KMAJOR) return *reinterpret_cast<const f32x4*>(slot + li * Panel<false>::LD + k + 4 * h);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = slot[(k + 4 * h + q) * 32 + li];
    return v;
}

__device__ __forceinline__ void epilogue_atomic(const GemmArgs& g, int row, int col, float v, bool lead) {
    if (row >= g.M || col >= g.N) return;
    v *= g.alpha;
    if (lead && g.bias) v += g.bias[col];
    atomicAdd(g.C + (size_t)row * g.ldc + col, v);
}

template <bool AKM, bool BKM>
__global__ void __launch_bounds__(256) gemm_panel_kernel(GemmArgs g, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FA = Panel<AKM>::FLOATS, FB = Panel<BKM>::FLOATS;
    float* As = smem;             // 2 slots
    float* Bs = smem + 2 * FA;    // 2 slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    int tm_, tn_; tile_coords(blockIdx.x, (g.M + 32 - 1) / 32, tiles_n, tm_, tn_);
    const int m0 = tm_ * 32, n0 = tn_ * 32;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int ns = g.K / KS / nsplit;              // stages handled by this workgroup
    const int kbase = split * ns * KS;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // prologue: up to two stages in flight
    panel_dma<AKM>(g.A, g.lda, m0, g.M, kbase, As, wave, lane);
    panel_dma<BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, kbase, Bs, wave, lane);
    if (ns > 1) {
        panel_dma<AKM>(g.A, g.lda, m0, g.M, kbase + KS, As + FA, wave, lane);
        panel_dma<BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, kbase + KS, Bs + FB, wave, lane);
    }
    for (int st = 0; st < ns; ++st) {
        const int cur = st & 1;
        // retire stage `st` (older than anything issued after it): leave the younger stage in flight
        if (st + 1 < ns) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* a_s = As + cur * FA;
        const float* b_s = Bs + cur * FB;
        const int kw = wave * (KS / 4);
#pragma unroll
        for (int g8 = 0; g8 < KS / 32; ++g8) {
            const f32x4 av = panel_frag<AKM>(a_s, li, kw + 8 * g8, h);
            const f32x4 bv = panel_frag<BKM>(b_s, li, kw + 8 * g8, h);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
        if (st + 2 < ns) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave finished reading slot `cur`
            panel_dma<AKM>(g.A, g.lda, m0, g.M, kbase + (st + 2) * KS, As + cur * FA, wave, lane);
            panel_dma<BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, kbase + (st + 2) * KS, Bs + cur * FB, wave, lane);
        }
    }
    // cross-wave reduction through LDS (slot memory is dead after this barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = smem;
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane] + red[(2 * 16 + r) * 64 + lane];
            if (nsplit > 1) epilogue_atomic(g, m0 + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + li, v, split == 0);
            else epilogue_store(g, m0 + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + li, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ring kernel: BMxBN output tile (64x64 with 128-deep stages, or 128x128 with 64-deep stages), 4 waves as 2x2, each
// wave a (BM/2)x(BN/2) block of 32x32 MFMA tiles over the FULL K (no in-workgroup reduction).  Operand stages arrive
// by LDS-DMA into a 2-slot ring exactly like the panel kernel (16 x 1-KiB global_load_lds per wave per stage, counted
// vmcnt, raw barriers), but a workgroup now moves (BM+BN)/(BM*BN) = 1/32 .. 1/64 operand bytes per MAC instead of 1/16:
// at ~30 GB/s of DMA ingest per CU that is what keeps the fp32 MFMA pipe (614 GFLOP/s per CU) fed.
// LDS image, k-contiguous operand: rows are stored in groups of RPI = 256/KS rows (one 1-KiB DMA each) + 4 floats pad;
// k-major operand: [k][rows] unpadded (b32 fragment reads across consecutive lanes, conflict-free).
// Requirements: K % KS == 0, 16-B aligned rows; k-major operands need rows % BM == 0.
// ------------------------------------------------------------------------------------------------
template <int ROWS, int KSTAGE, bool KMAJOR>
struct Ring {
    static constexpr int RPI = 256 / KSTAGE;                                   // rows per DMA instruction (k-contiguous)
    static constexpr int GROUP = RPI * KSTAGE + 4;                             // floats per row group incl. pad
    static constexpr int FLOATS = KMAJOR ? KSTAGE * ROWS : (ROWS / RPI) * GROUP;
    static constexpr int INSTR = ROWS * KSTAGE / 256;                          // DMA instructions per stage (all waves)
    __device__ static __forceinline__ int row_off(int r) { return (r / RPI) * GROUP + (r % RPI) * KSTAGE; }
};

template <int ROWS, int KSTAGE, bool KMAJOR, int AUX = 0>
__device__ __forceinline__ void ring_dma(const float* __restrict__ G, int ld, int r0, int R, int k0, float* slot, int wave, int lane) {
    using RG = Ring<ROWS, KSTAGE, KMAJOR>;
    constexpr int PER_WAVE = RG::INSTR / 4;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int ins = wave * PER_WAVE + j;
        const float* src; float* dst;
        if (!KMAJOR) {
            constexpr int LPR = KSTAGE / 4;                                    // lanes per row
            const int r = ins * RG::RPI + lane / LPR;
            int gr = r0 + r; gr = gr < R ? gr : R - 1;
            src = G + (size_t)gr * ld + k0 + 4 * (lane % LPR);
            dst = slot + (ins) * RG::GROUP;                                    // group `ins` = rows ins*RPI ..
        } else {
            constexpr int LPK = ROWS / 4;                                      // lanes per k-row
            constexpr int KPI = 64 / LPK;                                      // k-rows per instruction
            const int kb = ins * KPI;
            src = G + (size_t)(k0 + kb + lane / LPK) * ld + r0 + 4 * (lane % LPK);
            dst = slot + kb * ROWS;
        }
        __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)dst, 16, 0, AUX);
    }
}

template <int ROWS, int KSTAGE, bool KMAJOR>
__device__ __forceinline__ f32x4 ring_frag(const float* __restrict__ slot, int r, int k, int h) {
    using RG = Ring<ROWS, KSTAGE, KMAJOR>;
    if (!KMAJOR) return *reinterpret_cast<const f32x4*>(slot + RG::row_off(r) + k + 4 * h);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = slot[(k + 4 * h + q) * ROWS + r];
    return v;
}

template <int BM, int BN, int KSTAGE, bool AKM, bool BKM>
__global__ void __launch_bounds__(256) gemm_ring_kernel(GemmArgs g, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RA = Ring<BM, KSTAGE, AKM>; using RB = Ring<BN, KSTAGE, BKM>;
    constexpr int TM = BM / 64, TN = BN / 64;
    static_assert((RA::INSTR + RB::INSTR) / 4 == 16, "vmcnt(16) below assumes 16 DMA instructions per wave per stage");
    float* As = smem; float* Bs = smem + 2 * RA::FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    int tm_, tn_; tile_coords(blockIdx.x, (g.M + BM - 1) / BM, tiles_n, tm_, tn_);
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const int ns = g.K / KSTAGE;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    ring_dma<BM, KSTAGE, AKM>(g.A, g.lda, m0, g.M, 0, As, wave, lane);
    ring_dma<BN, KSTAGE, BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, 0, Bs, wave, lane);
    if (ns > 1) {
        ring_dma<BM, KSTAGE, AKM>(g.A, g.lda, m0, g.M, KSTAGE, As + RA::FLOATS, wave, lane);
        ring_dma<BN, KSTAGE, BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, KSTAGE, Bs + RB::FLOATS, wave, lane);
    }
    for (int st = 0; st < ns; ++st) {
        const int cur = st & 1;
        if (st + 1 < ns) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* a_s = As + cur * RA::FLOATS;
        const float* b_s = Bs + cur * RB::FLOATS;
#pragma unroll 1
        for (int g8 = 0; g8 < KSTAGE / 8; ++g8) {
            f32x4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = ring_frag<BM, KSTAGE, AKM>(a_s, wm + 32 * i + li, 8 * g8, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = ring_frag<BN, KSTAGE, BKM>(b_s, wn + 32 * j + li, 8 * g8, h);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
        }
        if (st + 2 < ns) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ring_dma<BM, KSTAGE, AKM>(g.A, g.lda, m0, g.M, (st + 2) * KSTAGE, As + cur * RA::FLOATS, wave, lane);
            ring_dma<BN, KSTAGE, BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, (st + 2) * KSTAGE, Bs + cur * RB::FLOATS, wave, lane);
        }
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
        for (int j = 0; j < TN; ++j)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r)
                epilogue_store(g, m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + wn + 32 * j + li, acc[i][j][r]);
}

// ------------------------------------------------------------------------------------------------
// Panel kernel with 128-deep stages: the same 32x32 tile / 4-way in-workgroup split-K / LDS-DMA structure at HALF the LDS
// (67 KB), so TWO workgroups share a CU.  Used when a small-M layer has more 32x32 tiles than the chip has CUs (the 768-row
// reverse loop: 384 tiles): with the 133 KB variant the second half of the grid waits for the first to retire.
// ------------------------------------------------------------------------------------------------
template <bool AKM, bool BKM, int KST>
__global__ void __launch_bounds__(256) gemm_panel128_kernel(GemmArgs g, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RA = Ring<32, KST, AKM>; using RB = Ring<32, KST, BKM>;
    static_assert((RA::INSTR + RB::INSTR) / 4 == KST / 16, "counted vmcnt below assumes KST/16 DMA instructions per wave per stage");
    float* As = smem; float* Bs = smem + 2 * RA::FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    int tm_, tn_; tile_coords(blockIdx.x, (g.M + 31) / 32, tiles_n, tm_, tn_);
    const int m0 = tm_ * 32, n0 = tn_ * 32;
    const int ns = g.K / KST;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ring_dma<32, KST, AKM>(g.A, g.lda, m0, g.M, 0, As, wave, lane);
    ring_dma<32, KST, BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, 0, Bs, wave, lane);
    if (ns > 1) {
        ring_dma<32, KST, AKM>(g.A, g.lda, m0, g.M, KST, As + RA::FLOATS, wave, lane);
        ring_dma<32, KST, BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, KST, Bs + RB::FLOATS, wave, lane);
    }
    for (int st = 0; st < ns; ++st) {
        const int cur = st & 1;
        if (st + 1 < ns) { if (KST == 128) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* a_s = As + cur * RA::FLOATS;
        const float* b_s = Bs + cur * RB::FLOATS;
        const int kw = wave * (KST / 4);
#pragma unroll
        for (int g8 = 0; g8 < KST / 32; ++g8) {
            const f32x4 av = ring_frag<32, KST, AKM>(a_s, li, kw + 8 * g8, h);
            const f32x4 bv = ring_frag<32, KST, BKM>(b_s, li, kw + 8 * g8, h);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
        if (st + 2 < ns) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ring_dma<32, KST, AKM>(g.A, g.lda, m0, g.M, (st + 2) * KST, As + cur * RA::FLOATS, wave, lane);
            ring_dma<32, KST, BKM, AID_PANEL_B_AUX>(g.B, g.ldb, n0, g.N, (st + 2) * KST, Bs + cur * RB::FLOATS, wave, lane);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = smem;
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane] + red[(2 * 16 + r) * 64 + lane];
            epilogue_store(g, m0 + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + li, v);
        }
    }
}
template <bool AKM, bool BKM, int KST = 128>
static int launch_panel128(const GemmArgs& g, hipStream_t s) {
    t_variant = KST == 64 ? 15 : 14;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    const size_t lds = sizeof(float) * 2 * (Ring<32, KST, AKM>::FLOATS + Ring<32, KST, BKM>::FLOATS);
    static bool done = false;
    if (!done && lds > 65536) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_panel128_kernel<AKM, BKM, KST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    hipLaunchKernelGGL((gemm_panel128_kernel<AKM, BKM, KST>), dim3(tm * tn), dim3(256), lds, s, g, tn);
    return AID_OK;
}

template <int BM, int BN, int KSTAGE, bool AKM, bool BKM>
static int launch_ring(const GemmArgs& g, hipStream_t s) {
    t_variant = 2;
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const size_t lds = sizeof(float) * 2 * (Ring<BM, KSTAGE, AKM>::FLOATS + Ring<BN, KSTAGE, BKM>::FLOATS);
    static bool done = false;
    if (!done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<BM, BN, KSTAGE, AKM, BKM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, KSTAGE, AKM, BKM>), dim3(tm * tn), dim3(256), lds, s, g, tn);
    return AID_OK;
}

template <bool AKM, bool BKM>
static int launch_panel(const GemmArgs& g, hipStream_t s) {
    t_variant = 1;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    const size_t lds = sizeof(float) * 2 * (Panel<AKM>::FLOATS + Panel<BKM>::FLOATS);
    static bool done = false;
    if (!done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_panel_kernel<AKM, BKM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    int ks = g.ksplit > 1 ? g.ksplit : 1;
    while (ks > 1 && (g.K / KS) % ks) --ks;
    hipLaunchKernelGGL((gemm_panel_kernel<AKM, BKM>), dim3(tm * tn, ks), dim3(256), lds, s, g, tn);
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool AKM, bool BKM>
static int launch_tile(const GemmArgs& g, bool vec, hipStream_t s) {
    t_variant = vec ? 4 : 5;
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const size_t lds = sizeof(float) * 2 * (Tile<BM, AKM>::FLOATS + Tile<BN, BKM>::FLOATS);
    if (lds > 65536) {  // opt in to >64 KiB dynamic LDS once per instantiation
        static bool done = false;
        if (!done) {
            AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tile_kernel<BM, BN, AKM, BKM, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tile_kernel<BM, BN, AKM, BKM, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done = true;
        }
    }
    const int ny = g.slab ? g.nslab : 1;
    if (vec) hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, AKM, BKM, true>), dim3(tm * tn, ny), dim3(256), lds, s, g, tn);
    else     hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, AKM, BKM, false>), dim3(tm * tn, ny), dim3(256), lds, s, g, tn);
    return AID_OK;
}
static int g_big_map = 0;     // benches only (aid_gemm_force_kernel bits 8+); the automatic dispatcher passes the map explicitly
template <int BM, int BN, int BKT, bool AKM, bool BKM, int PD = 1>
static int launch_big(const GemmArgs& g, hipStream_t s, int map = -1) {
    t_variant = PD == 2 ? 9 : 11;
    if (map < 0) map = g_big_map;
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const size_t lds = sizeof(float) * 2 * (TileK<BM, AKM, BKT>::FLOATS + TileK<BN, BKM, BKT>::FLOATS);
    static bool attr_done = false;
    if (lds > 65536 && !attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, BKT, AKM, BKM, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_big_kernel<BM, BN, BKT, AKM, BKM, PD>), dim3(tm * tn, g.slab ? g.nslab : 1), dim3(256), lds, s, g, tn, map);
    return AID_OK;
}
// deep-K slab launch (gemm_big_kernel with FL): 128 x 128 tiles from 128 rows, 64 x 64 below; 1-D grid of tiles x slices
constexpr int DEEPK_MIN = 65536, DEEPK_FL = 2;
template <int BM, int BN>
static int launch_deepk(const GemmArgs& g, hipStream_t s) {
    t_variant = 12;
    constexpr int BKT = 32;
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const size_t lds = sizeof(float) * 2 * (TileK<BM, false, BKT>::FLOATS + TileK<BN, false, BKT>::FLOATS);
    static bool attr_done = false;
    if (lds > 65536 && !attr_done) {
        AID_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, BKT, false, false, 2, DEEPK_FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_big_kernel<BM, BN, BKT, false, false, 2, DEEPK_FL>), dim3(tm * tn * g.nslab), dim3(256), lds, s, g, tn, 0);
    return AID_OK;
}
template <bool AKM, bool BKM>
static int launch_splitk(const GemmArgs& g, bool vec, hipStream_t s) {
    t_variant = vec ? 6 : 7;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    const size_t lds = sizeof(float) * (2 * (Tile<32, AKM>::FLOATS + Tile<32, BKM>::FLOATS) + 3 * 16 * 64);
    if (vec) hipLaunchKernelGGL((gemm_splitk_kernel<AKM, BKM, true>), dim3(tm * tn), dim3(256), lds, s, g, tn);
    else     hipLaunchKernelGGL((gemm_splitk_kernel<AKM, BKM, false>), dim3(tm * tn), dim3(256), lds, s, g, tn);
    return AID_OK;
}

static int launch_skinny(const GemmArgs& g, hipStream_t s) {
    t_variant = 10;
    const int kv = (g.K + 255) / 256;
    if (kv <= 1) return launch_skinny_kv<1>(g, s);
    if (kv <= 2) return launch_skinny_kv<2>(g, s);
    if (kv <= 4) return launch_skinny_kv<4>(g, s);
    if (kv <= 8) return launch_skinny_kv<8>(g, s);
    return launch_skinny_kv<16>(g, s);
}
static int launch_thin_tn(const GemmArgs& g, hipStream_t s) {
    t_variant = 8;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    hipLaunchKernelGGL(gemm_thin_tn_kernel, dim3(tm * tn), dim3(1024), 0, s, g, tn);
    return AID_OK;
}
// Dispatch hints are a property of the STREAM a launch goes to (aid_stream_hint), never of the process: a side stream whose small
// GEMMs must co-reside with another stream's big ones is marked once, and a thread launching on any other stream (the collector's
// inference thread of utils/async_collector.py:388,451) is unaffected by it.  Lock-free readers; writers serialise on a mutex.
struct StreamHint { std::atomic<void*> stream{nullptr}; std::atomic<int> flags{0}; std::atomic<int> used{0}; };
static StreamHint g_stream_hints[64];
static std::mutex g_stream_hints_mu;
int stream_hint_set(hipStream_t s, int flags) {
    std::lock_guard<std::mutex> lk(g_stream_hints_mu);
    int free_slot = -1;
    for (int i = 0; i < 64; ++i) {
        if (g_stream_hints[i].used.load(std::memory_order_acquire)) {
            if (g_stream_hints[i].stream.load(std::memory_order_relaxed) == (void*)s) {
                g_stream_hints[i].flags.store(flags, std::memory_order_release);
                if (flags == 0) g_stream_hints[i].used.store(0, std::memory_order_release);      // cleared: the slot is free again
                return AID_OK;
            }
        } else if (free_slot < 0) free_slot = i;
    }
    if (flags == 0) return AID_OK;                 // nothing recorded for this stream and nothing to record
    AID_REQUIRE(free_slot >= 0, AID_E_SHAPE, "aid_stream_hint: all 64 slots hold live hints (clear a stream's flags to free its slot)");
    g_stream_hints[free_slot].stream.store((void*)s, std::memory_order_relaxed);
    g_stream_hints[free_slot].flags.store(flags, std::memory_order_relaxed);
    g_stream_hints[free_slot].used.store(1, std::memory_order_release);
    return AID_OK;
}
static inline int stream_hint_get(hipStream_t s) {
    for (int i = 0; i < 64; ++i) {
        if (!g_stream_hints[i].used.load(std::memory_order_acquire)) continue;
        if (g_stream_hints[i].stream.load(std::memory_order_relaxed) == (void*)s) return g_stream_hints[i].flags.load(std::memory_order_acquire);
    }
    return 0;
}
int stream_hint_flags(hipStream_t s) { return stream_hint_get(s); }
// TESTS / BENCHES ONLY (aid_gemm_force_kernel): a process-wide kernel override, deliberately not part of the product path and not thread-safe.
static int g_force_kernel = 0;   // 0 auto, 1 panel32, 2 ring64, 4 tile64, 6 splitk, 9 big (PD 2), 11 big (PD 1), 14 panel128

template <bool AKM, bool BKM>
static int dispatch(const GemmArgs& g, bool vec, hipStream_t s) {
    const long t32 = (long)((g.M + 31) / 32) * ((g.N + 31) / 32);
    const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
    const bool r64_ok = vec && g.K >= 128 && g.K % 128 == 0 && (!AKM || g.M % 64 == 0) && (!BKM || g.N % 64 == 0);
    const bool p32_ok = vec && g.K >= KS && g.K % KS == 0 && (!AKM || g.M % 32 == 0) && (!BKM || g.N % 32 == 0);
    switch (g_force_kernel) {          // benches / tests only
        case 1: if (p32_ok) return launch_panel<AKM, BKM>(g, s); break;
        case 2: if (r64_ok) return launch_ring<64, 64, 128, AKM, BKM>(g, s); break;
        case 4: return launch_tile<64, 64, AKM, BKM>(g, vec, s);
        case 6: return launch_splitk<AKM, BKM>(g, vec, s);
        case 8: if (AKM && BKM) return launch_thin_tn(g, s); break;
        case 10: if (!AKM && !BKM && vec && g.K >= 4 && g.K <= 4096) return launch_skinny(g, s); break;
        case 9: if (vec) return launch_big<64, 64, 32, AKM, BKM, 2>(g, s); break;
        case 11: if (vec) return launch_big<64, 64, 32, AKM, BKM, 1>(g, s); break;
        case 12: if (vec) return launch_big<128, 128, 32, AKM, BKM, 2>(g, s); break;
        case 14: if (p32_ok) return launch_panel128<AKM, BKM>(g, s); break;
        case 15: if (p32_ok) return launch_panel128<AKM, BKM, 64>(g, s); break;
        default: break;
    }
    // Automatic selection, set by measurement on MI355X (profiles/gemm_*_r01.txt, tools/perf/cold_chain.py).
    // K <= 32 (first layer of every network, K = latent / observation dim): VALU kernel, output-write bound.
    if (!AKM && g.K >= 1 && g.K <= 32 && (!BKM || g.M >= 64)) return launch_smallk(g, s);
    // act() / collector batches: one wave per output column, W row in registers (weight-streaming, not MFMA-shaped)
    if (!AKM && !BKM && vec && g.M <= 32 && g.K >= 64 && g.K <= 4096 && g.ksplit <= 1) return launch_skinny(g, s);
    // one column tile under many row tiles (N = 17 / 6 / 1 heads over 25 600 rows): an A-streaming problem; the 35 KB panel variant keeps
    // 4 workgroups x 64-deep stages per CU in flight (18.9 vs 27.7 us on 25 600 x 17 x 512)
    if (p32_ok && g.N <= 32 && t64 >= 160 && g.ksplit <= 1) return launch_panel128<AKM, BKM, 64>(g, s);
    if (vec && g.K >= 64) {
        // >= 2048 tiles (the 25 600-row MINE / hoisted-conditioning GEMMs): 64x64x32 tile at 4 workgroups/CU, branch-free loads, two
        // tiles of prefetch, XCD-grouped tile map (+3..20 % on these shapes); 160..2047 tiles: the same kernel, plain tile map.
        if (t64 >= 2048) return launch_big<64, 64, 32, AKM, BKM, 2>(g, s, 1);
        // exactly one 64x64 tile per CU with K >= 512 (M = 512 x N = 2048 in a 2B-row reverse loop): the LDS-DMA ring keeps two
        // 128-deep stages in flight and is 2.4 us per launch faster in situ (cold weights)
        if (t64 >= 224 && t64 <= 288 && r64_ok && g.K >= 512) return launch_ring<64, 64, 128, AKM, BKM>(g, s);
        if (t64 >= 160) return launch_big<64, 64, 32, AKM, BKM, 2>(g, s, 0);
    }
    if (t64 >= 160) return launch_tile<64, 64, AKM, BKM>(g, vec, s);        // unaligned operands / K < 64
    // < 160 tiles: the grid cannot fill the chip with 64x64 tiles; the 32x32 LDS-DMA panel kernel (whole K in flight, 4-way
    // in-workgroup split-K) has the shortest critical path.
    if (p32_ok && t32 <= 1536 && (stream_hint_get(s) & 1)) return launch_panel128<AKM, BKM, 64>(g, s);      // side-stream launches: co-reside with the big GEMM
    if (p32_ok && t32 <= 1536) {
        // more tiles than CUs and >= 16 row tiles (the 768-row reverse loop, 1024-row batches): two 67 KB workgroups per CU beat two
        // rounds of the 133 KB kernel by 12-18 us per launch in situ; a 256-row x 2048-column layer is the exception
        if (t32 > 272 && g.M >= 512 && g.ksplit <= 1) return launch_panel128<AKM, BKM>(g, s);
        return launch_panel<AKM, BKM>(g, s);
    }
    if (g.K <= 64) return launch_tile<64, 64, AKM, BKM>(g, vec, s);
    // a few output tiles under a long K (weight gradients of 6- / 17-wide layers): 16 waves split K inside one workgroup
    if (AKM && BKM && t32 <= 64 && g.K >= 512) return launch_thin_tn(g, s);
    return launch_splitk<AKM, BKM>(g, vec, s);
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool mid_route_ok(const GemmArgs& g, bool vec) {
    if (!vec || g.a_kmajor || g.mask || g.accumulate || g.slab || g.ksplit > 1 || g.alpha != 1.f) return false;
    if (g.b_kmajor && (g.act != ACT_NONE || g.N % 64 != 0)) return false;
    if (!gemm_mid_ok(g.M, g.N, g.K, 1, g.act, g.A, g.lda, g.B, g.ldb) || g.N % 64 != 0) return false;
    if (g.M > 1024) { const long wg = (long)(g.M / 80) * (g.N / 64); return g.M % 80 == 0 && wg >= 192 && wg <= 512; }
    // 8 row tiles x >= 8 column tiles: even as 64 workgroups (N = 512, K = 512: 8.8 us) the eight-stage pipeline beats the 32 x 32 panel kernel's
    // 256 workgroups (11 - 13 us) on the ELBO passes' 256- / 512-row products
    return g.N >= 512 && g.N <= 4096;
}
static thread_local SplitScope* t_split = nullptr;
SplitScope::SplitScope(Workspace& ws, hipStream_t st, size_t fl, int nc) : slab(nullptr), floats(fl), ctr(nullptr), nctr(nc), s(st), prev(t_split) {
    // The scratch is optional: a workspace sized without it (a caller holding the previous formula's size) must degrade to "no route", not to
    // AID_E_WORKSPACE - so look before allocating (Workspace::raw marks an overflow that the next AID_WS_OK would report).
    const size_t a0 = (ws.off + 255) & ~(size_t)255, a1 = (a0 + fl * sizeof(float) + 255) & ~(size_t)255;
    if (ws.base != nullptr && !ws.overflow && a1 + nc * sizeof(int) <= ws.bytes) {
        slab = ws.f32(fl); ctr = (int*)ws.raw(nc * sizeof(int));
        if (!slab || !ctr || hipMemsetAsync(ctr, 0, nc * sizeof(int), st) != hipSuccess) { slab = nullptr; ctr = nullptr; }      // no scratch: no route
    }
    t_split = this;
}
SplitScope::~SplitScope() { t_split = prev; }
// K slices for the split route: 0 = not this problem.  8 row tiles x N / 64 column tiles; slices double while they still fit the chip and
// leave each an even number of 64-deep stages.
static inline int mid_split_factor(const GemmArgs& g, bool vec, hipStream_t s) {
    const SplitScope* sc = t_split;
    if (!sc || !sc->slab || sc->s != s || !vec || g.a_kmajor || g.mask || g.slab || g.ksplit > 1 || g.alpha != 1.f) return 0;
    // (K = 512: two slices of four stages each measured level with the panel kernel at 256 rows and behind it at 512 - the fold's tail, stores
    // through the L2, the counter and the read-back, is ~4 us)
    if (g.M <= 128 || g.M > 1024 || g.N % 64 != 0 || g.N > 1024 || g.K < 1024 || (g.b_kmajor && g.act != ACT_NONE)) return 0;
    if (!(g.act == ACT_NONE || g.act == ACT_GELU || g.act == ACT_SILU || g.act == ACT_RELU)) return 0;
    const int tiles = 8 * (g.N / 64);
    int ks = 1;
    while (ks < 8 && tiles * ks * 2 <= 256 && g.K % (256 * ks) == 0 && g.K / (2 * ks) >= 256) ks *= 2;      // a slice is at least four 64-deep stages
    if (ks < 2 || tiles > sc->nctr || (size_t)ks * g.M * g.N > sc->floats) return 0;
    return gemm_mid_ok(g.M, g.N, g.K, ks, g.act, g.A, g.lda, g.B, g.ldb) ? ks : 0;
}
// true when the automatic dispatcher sends this problem to gemm_tile_kernel (profiling label only)
static inline bool routes_to_tile(const GemmArgs& g) {
    return g_force_kernel == 0 && (long)((g.M + 63) / 64) * ((g.N + 63) / 64) >= 160;
}

void gemm_force_kernel(int k) { g_force_kernel = k & 0xff; g_big_map = (k >> 8) & 1; }

__global__ void slab_fold_kernel(const float* __restrict__ slab, int nslab, GemmArgs g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t mn = (size_t)g.M * g.N;
    if (i >= mn) return;
    double vd = 0.0;                      // the slices are folded in float64, in slice order: the fold adds no rounding of its own (HBM-bound: free)
    for (int sl = 0; sl < nslab; ++sl) vd += (double)slab[(size_t)sl * mn + i];
    float v = (float)vd;
    const int row = i / g.N, col = i % g.N;
    v *= g.alpha;
    if (g.bias) v += g.bias[col];
    const size_t o = (size_t)row * g.ldc + col;
    if (g.accumulate) v += g.C[o];
    g.C[o] = v;
}

int gemm_f32(const GemmArgs& g, hipStream_t s) {
    AID_REQUIRE(g.A && g.B && g.C, AID_E_NULL, "gemm_f32: null operand");
    AID_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0, AID_E_SHAPE, "gemm_f32: negative dim");
    AID_REQUIRE(!(g.accumulate && g.act != ACT_NONE), AID_E_SHAPE, "gemm_f32: accumulate with activation");
    AID_REQUIRE(g.ksplit <= 1 || (g.act == ACT_NONE && !g.preact && !g.mask && !g.dact_pre && !g.residual && !g.accumulate), AID_E_SHAPE,
                "gemm_f32: ksplit needs a plain epilogue (C pre-initialised by the caller)");
    if (g.M == 0 || g.N == 0) return AID_OK;
    // vector (16-B) global loads need every row start 16-B aligned along the contiguous dim
    bool vec = aligned16(g.A) && aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0);
    vec = vec && (g.a_kmajor ? (g.M % 4 == 0) : (g.K % 4 == 0)) && (g.b_kmajor ? (g.N % 4 == 0) : (g.K % 4 == 0));
    if (g.slab) {   // deterministic split-K: K-slices -> slab -> ordered fold
        AID_REQUIRE(g.nslab >= 1 && g.act == ACT_NONE && !g.preact && !g.mask && !g.dact_pre && !g.residual, AID_E_SHAPE, "gemm_f32: slab mode needs a plain epilogue");
        ProfScope prof(PROF_GEMM_TILE, s, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
        prof.m = g.M; prof.n = g.N; prof.k = g.K; prof.layout = g.a_kmajor * 2 + g.b_kmajor + 4;
        int r;
        if (vec && !g.a_kmajor && !g.b_kmajor && g.K >= DEEPK_MIN)      // blocked float32 accumulation, XCD-grouped slices (the encoder head's K = 451 584)
            r = g.M >= 128 ? launch_deepk<128, 128>(g, s) : launch_deepk<64, 64>(g, s);
        else if (vec && !g.a_kmajor && !g.b_kmajor) r = launch_big<64, 64, 32, false, false, 2>(g, s, 0);      // branch-free, two tiles of prefetch
        else if (!g.a_kmajor && !g.b_kmajor) r = launch_tile<64, 64, false, false>(g, vec, s);
        else if (!g.a_kmajor && g.b_kmajor) r = launch_tile<64, 64, false, true>(g, vec, s);
        else if (g.a_kmajor && g.b_kmajor) r = launch_tile<64, 64, true, true>(g, vec, s);
        else r = launch_tile<64, 64, true, false>(g, vec, s);
        prof.variant = t_variant;
        if (r != AID_OK) return r;
        const size_t mn = (size_t)g.M * g.N;
        hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, g.slab, g.nslab, g);
        AID_LAUNCH_CHECK();
        return AID_OK;
    }
    ProfScope prof(routes_to_tile(g) ? PROF_GEMM_TILE : PROF_GEMM, s, 2.0 * g.M * g.N * g.K,
                   4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    // tall Linear forward with a 512-multiple width and a plain epilogue: 64-row x 512-column full-K workgroups (gemm_rowln.hip) - per
    // barrier four times the MFMA work of the 64x64 tile, hand-over hidden in the MFMA stream (25 600 x 512 x 512: 108 vs 84 TFLOP/s)
    if (g_force_kernel == 0 && !g.a_kmajor && !g.b_kmajor && !g.preact && !g.dact_pre && !g.accumulate && !g.slab && g.ksplit <= 1 && g.alpha == 1.f &&
        linear_rows_ok(g.M, g.N, g.K, g.act, g.A, g.lda, g.B, g.ldb) ) {
        const bool d16 = g.K % 16 == 0 && (stream_hint_get(s) & 2);
        prof.m = g.M; prof.n = g.N; prof.k = g.K; prof.layout = g.act + 8 * (g.K % (d16 ? 16 : 32) != 0) + 32 * d16; prof.variant = 23; prof.id = PROF_GEMM_TILE;
        if (rowln_x6_enabled() && rowln_x6_ok(g.N, g.K, false)) { prof.variant = 29; prof.layout = g.act; }      // the six-term bf16 form (gemm_rowln_x6.hip)
        return linear_rows(g.A, g.lda, g.B, g.ldb, g.bias, g.act, g.mask, g.ldc, g.residual, g.ldr, g.C, g.ldc, g.M, g.N, g.K, s);
    }
    // layers that one-workgroup-per-CU tiles fit exactly (gemm_mid.hip): K*B = 2 560 rows x 512 columns of the rollout / BPTT as 80 x 64 tiles,
    // 129..1024 rows x >= 2048 columns (the DiT MLP's wide layer in the ELBO passes) as 8 row tiles; forward (W [N][K]) and data-gradient
    // (W [K][N]) forms, optional pre-activation output.  The profiler records them under gemm_mid's own entry.
    if (g_force_kernel == 0 && gemm_thin_n_ok(g)) {
        prof.m = g.M; prof.n = g.N; prof.k = g.K; prof.layout = g.b_kmajor; prof.variant = 25; prof.id = PROF_GEMM;
        return gemm_thin_n(g, s);
    }
    if (const int ks = g_force_kernel == 0 ? mid_split_factor(g, vec, s) : 0) {
        prof.cancel();
        return gemm_mid_ex(g.A, g.lda, g.B, g.ldb, g.b_kmajor, g.bias, g.act, g.residual, g.ldr, g.C, g.ldc, g.preact, t_split->slab, g.M, g.N, g.K, ks, s, g.dact_pre,
                           g.dact, t_split->ctr, g.accumulate);
    }
    if (g_force_kernel == 0 && mid_route_ok(g, vec)) {
        prof.cancel();
        return gemm_mid_ex(g.A, g.lda, g.B, g.ldb, g.b_kmajor, g.bias, g.act, g.residual, g.ldr, g.C, g.ldc, g.preact, nullptr, g.M, g.N, g.K, 1, s, g.dact_pre, g.dact);
    }
    prof.m = g.M; prof.n = g.N; prof.k = g.K; prof.layout = g.a_kmajor * 2 + g.b_kmajor;
    int r;
    if (!g.a_kmajor && !g.b_kmajor) r = dispatch<false, false>(g, vec, s);
    else if (!g.a_kmajor && g.b_kmajor) r = dispatch<false, true>(g, vec, s);
    else if (g.a_kmajor && g.b_kmajor) r = dispatch<true, true>(g, vec, s);
    else r = dispatch<true, false>(g, vec, s);
    prof.variant = t_variant;
    prof.id = (t_variant == 9 || t_variant == 11 || t_variant == 4 || t_variant == 5) ? PROF_GEMM_TILE : PROF_GEMM;   // family = the kernel that ran
    if (r != AID_OK) return r;
    AID_LAUNCH_CHECK();
    return AID_OK;
}

}  // namespace aid
