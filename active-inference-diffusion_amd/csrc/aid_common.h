// Internal helpers shared by every translation unit of libaid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "../../include/aid_hip.h"

namespace aid {

// ---- error plumbing (never throw / abort across the C ABI) -------------------------------------
void set_error(const char* fmt, ...);
#define AID_CHECK_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::aid::set_error("%s:%d HIP error %d (%s) in `%s`", __FILE__, __LINE__, (int)_e, \
                             hipGetErrorString(_e), #expr);                                  \
            return AID_E_HIP;                                                                \
        }                                                                                    \
    } while (0)
#define AID_REQUIRE(cond, code, ...)         \
    do {                                     \
        if (!(cond)) {                       \
            ::aid::set_error(__VA_ARGS__);   \
            return (code);                   \
        }                                    \
    } while (0)
#define AID_TRY(expr)              \
    do {                           \
        int _r = (expr);           \
        if (_r != AID_OK) return _r; \
    } while (0)
#define AID_LAUNCH_CHECK() AID_CHECK_HIP(hipGetLastError())

// ---- caller-owned workspace: bump allocator, 256-B aligned carve-outs -------------------------
struct Workspace {
    char* base; size_t bytes; size_t off; bool overflow;
    Workspace(void* p, size_t n) : base((char*)p), bytes(n), off(0), overflow(false) {}
    float* f32(size_t n) { return (float*)raw(n * sizeof(float)); }
    double* f64(size_t n) { return (double*)raw(n * sizeof(double)); }
    void* raw(size_t n) {
        size_t a = (off + 255) & ~(size_t)255;
        if (base == nullptr || a + n > bytes) { overflow = true; off = a + n; return nullptr; }
        off = a + n; return base + a;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};
#define AID_WS_OK(ws)                                                                        \
    AID_REQUIRE(!(ws).overflow, AID_E_WORKSPACE, "workspace too small: need >= %zu bytes, have %zu", \
                (ws).off, (ws).bytes)

// ---- activation ids (GEMM epilogue + row-wise kernels) -----------------------------------------
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GELU = 3, ACT_MISH = 4, ACT_TANH = 5 };

// ---- GEMM:  C[M,N] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) (+ residual) (+ beta*C) ---------
// a_kmajor = 0: A stored [M][K] (k contiguous, lda = row stride);  1: stored [K][M] (m contiguous)
// b_kmajor = 0: B stored [N][K] (torch Linear weight);             1: stored [K][N]
struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K; int lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    const float* bias;       // [N] or null
    const float* residual;   // [M,N], row stride ldr, or null (added after the activation)
    int ldr;
    float* preact;           // optional second output: value before the activation (row stride ldc)
    int act;                 // Act
    float alpha;             // scale of the product
    int accumulate;          // 1: C += result (act must be NONE)
    const float* mask;       // optional [M,N] multiplicative mask applied after act (dropout), stride ldc
    const float* dact_pre;   // optional [M,N] (stride ldc): multiply by act'(dact_pre) of kind `dact` (tangent /
    int dact;                //   backward through an activation whose pre-activation was saved)
    float* slab;             // non-null: deterministic split-K for tiny-output / huge-K contractions (conv weight gradients, the
    int nslab;               //   451 584-wide Linear): the 64x64 tile kernel runs `nslab` K-slices per tile, each storing its raw partial
                             //   tile into slab[slice][M][N]; gemm_f32 then folds the slices in order (+bias, +C if accumulate)
    int ksplit;              // >1: panel kernel splits K over `ksplit` workgroups which atomically ADD into C (C must already hold
                             //     the residual / initial value; bias added by split 0; act must be NONE).  Order of the float adds,
                             //     hence the last bit, is not reproducible run to run.
};
inline GemmArgs gemm_args(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K) {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = 1.f;
    return g;
}
int gemm_f32(const GemmArgs& g, hipStream_t s);
// Scratch for gemm_f32's split-K route (129..1024 rows x <= 1024 columns: too few 64-wide tiles for 256 CUs, so K is cut into slices and the
// last slice to arrive folds them, gemm_mid.hip).  An entry point that owns a Workspace opens one scope for its stream; every gemm_f32 call of
// this thread on that stream may then take the route.  Slices and tile counters live in the caller's workspace: nothing is kept between calls.
struct SplitScope {
    float* slab; size_t floats; int* ctr; int nctr; hipStream_t s; SplitScope* prev;
    SplitScope(Workspace& ws, hipStream_t s, size_t floats = (size_t)1 << 20, int nctr = 256);
    ~SplitScope();
    static size_t bytes(size_t floats = (size_t)1 << 20, int nctr = 256) { return floats * sizeof(float) + nctr * sizeof(int) + 512; }
};
// thin outputs (N <= 32) at 33..8192 rows: W in LDS, a wave per four rows (gemm_thin.hip); gemm_f32 routes to it
bool gemm_thin_n_ok(const GemmArgs& g);
int gemm_thin_n(const GemmArgs& g, hipStream_t s);
// tall plain Linear through 64-row x 512-column full-K workgroups (gemm_rowln.hip); gemm_f32 routes to it
bool linear_rows_ok(int M, int N, int K, int act, const float* x, int ldx, const float* W, int ldw);
int linear_rows(const float* x, int ldx, const float* W, int ldw, const float* b, int act, const float* mask, int ldmask, const float* add, int ldadd,
                float* y, int ldy, int M, int N, int K, hipStream_t s);
void gemm_force_kernel(int k);   // tests / benches only: process-wide kernel override, 0 = automatic selection
int stream_hint_set(hipStream_t s, int flags);   // per-stream dispatch hints (bit 0: prefer low-LDS small-problem kernels; bit 1: 512-wide full-row kernels
                                                 // take 16-deep stages = 74 KB of LDS, so that another stream's 80 KB workgroups can share the CU)
int stream_hint_flags(hipStream_t s);            // lock-free read of a stream's flags (0 when none were set)
// y = x W^T + b with optional activation: the torch.nn.Linear forward.
inline int linear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int M, int N, int K,
                      int act, hipStream_t s, float* preact = nullptr, const float* residual = nullptr, int ldr = 0,
                      const float* mask = nullptr) {
    GemmArgs g = gemm_args(x, ldx, W, K, y, ldy, M, N, K);
    g.bias = b; g.act = act; g.preact = preact; g.residual = residual; g.ldr = ldr; g.mask = mask;
    return gemm_f32(g, s);
}
// dX[M,K] (+)= dY[M,N] W[N,K]
inline int linear_bwd_data(const float* dy, int lddy, const float* W, int K, float* dx, int lddx, int M, int N,
                           hipStream_t s, int accumulate = 0) {
    GemmArgs g = gemm_args(dy, lddy, W, K, dx, lddx, M, K, N);
    g.b_kmajor = 1; g.accumulate = accumulate;
    return gemm_f32(g, s);
}
// dW[N,K] (+)= dY[M,N]^T X[M,K]
inline int linear_bwd_weight(const float* dy, int lddy, const float* x, int ldx, float* dW, int M, int N, int K,
                             hipStream_t s, int accumulate = 1) {
    GemmArgs g = gemm_args(dy, lddy, x, ldx, dW, K, N, K, M);
    g.a_kmajor = 1; g.b_kmajor = 1; g.accumulate = accumulate;
    return gemm_f32(g, s);
}

// ---- grouped weight gradients (gemm_group.hip): the dW / db leaves of a backward pass are recorded while the adjoint chain runs and computed
// by ONE launch at the end of the pass.  The operands a recorded problem points at must stay untouched until flush().
constexpr int GROUP_MAX_SEG = 2, GROUP_MAX_PROB = 31;                 // 31 x 128 B + 8 B: the table travels in the kernel arguments (< 4 KB)
struct GroupSeg { const float* a; const float* b; int lda, ldb, m, _pad; };              // a = dY [m, N], b = X [m, K]
// C [N, K] += sum_s a_s^T b_s.  split > 1 (one-segment problems with a long contraction): the rows are cut into `split` chunks, each chunk's
// workgroups write raw partial tiles to part [split][N][K] (+ bias partials behind them) and a second small launch folds them in chunk order.
struct GroupProb { float* C; float* db; float* part; int ldc, N, K, nseg, tile0, tiles_k, vec, split, tiles_n, _pad; GroupSeg seg[GROUP_MAX_SEG]; };
struct GroupArgs { int nprob, _pad; GroupProb p[GROUP_MAX_PROB]; };
struct GradQueue {
    GroupArgs args; hipStream_t s; double flop;
    float* scratch = nullptr; size_t scratch_floats = 0, scratch_used = 0;       // optional, caller-owned: room for the partial tiles of split problems
    explicit GradQueue(hipStream_t st) : s(st), flop(0.0) { args.nprob = 0; args._pad = 0; }
    // dW[N, K] (row stride ldc) += dy^T x over M rows; db[N] += colsum(dy) when db != null.  A call with the same dW as the previous one
    // (and no db) becomes a second segment of that problem.
    int add(float* dW, int ldc, int N, int K, const float* dy, int lddy, const float* x, int ldx, int M, float* db);
    int flush();
};

// ---- profiling hooks (bench.py reads them through aid_prof_*) ---------------------------------
enum ProfKernel : int { PROF_GEMM = 0, PROF_ROWWISE = 1, PROF_OPTIM = 2, PROF_GEMM_TILE = 3, PROF_COUNT = 4 };
struct ProfScope {
    int id; hipStream_t s; bool on; hipEvent_t e0, e1; double flops, bytes; int m = 0, n = 0, k = 0, layout = 0, variant = 0;
    ProfScope(int id, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    void cancel();          // drop this record (the launch is accounted for by a nested scope)
};

// tanh(softplus(x)) = n / (n + 2) with n = e^x (e^x + 2): the same function as tanhf(log1pf(expf(x))) with one exponential and one division
// instead of three transcendental calls (the GroupNorm-Mish kernels of the pixel encoder are VALU-bound on them); tanh is 1 to fp32 precision
// beyond x = 20 (F.softplus switches to the identity there too) and e^{2x} would overflow beyond x = 44.
__device__ __forceinline__ float tanh_softplus(float x) {
    if (x > 20.f) return 1.f;
    const float e = expf(x), n = e * (e + 2.f);
    return n / (n + 2.f);
}
// e^x as one multiply and one v_exp_f32: the rounding of x log2(e) costs |x| 2^-24 of relative error (6e-7 at |x| = 10) against expf's
// 1 ulp, for 2 instructions instead of 12 (range reduction, ldexp, overflow selects).  Used where the exponential feeds a sigmoid, whose
// LayerNorm-ed argument is O(1) and whose result saturates long before the error grows; the full-row epilogues are VALU-bound on it.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_SILU: return x * __builtin_amdgcn_rcpf(1.f + exp_fast(-x));  // v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case ACT_MISH: return x * tanh_softplus(x);            // F.mish
        case ACT_TANH: return tanhf(x);
        default: return x;
    }
}
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + exp_fast(-x)); }
// first / second derivative of the activations (SiLU, exact-erf GELU, ReLU)
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_SILU: { const float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }
        case ACT_GELU: return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f;
        case ACT_MISH: { const float t = tanh_softplus(x); return t + x * sigmoidf_(x) * (1.f - t * t); }
        case ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
        default: return 1.f;
    }
}
__device__ __forceinline__ float act_grad2(float x, int act) {
    switch (act) {
        case ACT_SILU: { const float s = sigmoidf_(x); const float ds = s * (1.f - s); return ds * (2.f + x * (1.f - 2.f * s)); }
        case ACT_GELU: return expf(-0.5f * x * x) * 0.3989422804014327f * (2.f - x * x);
        default: return 0.f;
    }
}
// Sum over the 64 lanes of a (fully active) wave, the same value returned to every lane.  Data-parallel-primitive form: four adds inside a
// row of 16 (quad swaps, half-row and row mirrors), row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3, total read from lane 63 -
// ~12 VALU instructions where six `__shfl_xor` steps are six dependent trips through the LDS crossbar (ds_bpermute, ~100 cycles each).
// LayerNorm rows do two of these back to back; the 5-us row kernels of the 256-row chains are mostly that latency.  Fixed order: reproducible.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_add_(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add_<0xB1, 0xF, true>(v);        // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E, 0xF, true>(v);        // quad_perm [2,3,0,1]
    v = dpp_add_<0x141, 0xF, true>(v);       // row_half_mirror
    v = dpp_add_<0x140, 0xF, true>(v);       // row_mirror: every lane of a row holds the row's sum
    v = dpp_add_<0x142, 0xA, false>(v);      // row_bcast:15 -> rows 1 and 3
    v = dpp_add_<0x143, 0xC, false>(v);      // row_bcast:31 -> rows 2 and 3: lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// 8 per-lane partial sums -> 8 wave totals with 4 + 2 + 1 + 3 = 10 cross-lane moves (each butterfly stage halves the values a lane
// carries); afterwards lane 8 c (c = 0..7) holds the total of value c.  Fixed order: reproducible.
__device__ __forceinline__ float fold8(const float (&acc)[8], int lane) {
    float a4[4], a2[2], a1;
    const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float recv = __shfl_xor(h32 ? acc[i] : acc[i + 4], 32, 64); a4[i] = (h32 ? acc[i + 4] : acc[i]) + recv; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const float recv = __shfl_xor(h16 ? a4[i] : a4[i + 2], 16, 64); a2[i] = (h16 ? a4[i + 2] : a4[i]) + recv; }
    // the partner at distance 8 is inside the row of 16 (row_ror:8), and the last three steps are plain sums over 8 lanes: DPP, no LDS trip
    { const float send = h8 ? a2[0] : a2[1]; const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xF, 0xF, true)); a1 = (h8 ? a2[1] : a2[0]) + recv; }
    a1 = dpp_add_<0xB1, 0xF, true>(a1); a1 = dpp_add_<0x4E, 0xF, true>(a1); a1 = dpp_add_<0x141, 0xF, true>(a1);
    return a1;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace aid
