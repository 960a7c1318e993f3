// Grouped weight-gradient GEMM: every dW (and db) of one backward pass in ONE launch.
//
// Where it runs: the parameter gradients of the score network's backward passes (reference: autograd over models/score_networks.py inside
// core/active_inference.py:533-636, 709-729) and of the policy BPTT (core/active_inference.py:337-380).  A backward pass at 256 rows is a
// chain of ~40 Linear layers; the adjoint chain (mu_in = mu W) is sequential, but the weight gradients dW += mu^T x hang off it as leaves:
// nothing downstream reads them until the optimiser step.  Launched one by one inside the chain they are 64..256-tile problems that each
// cost a full kernel boundary (~17 us in situ: 100 of the 323 GEMM launches of a cfg-2 ELBO step, 45 of the 234 of the rollout's BPTT).
// Here the chain only RECORDS them (GradQueue::add) and a single launch at the end of the pass computes all of them: ~2 000-4 000 output
// tiles at once, which is the first time these layers fill 256 CUs.
//
// Problem p:  C_p[N_p, K_p] += sum over segments s of  A_s^T B_s,   A_s = dY [m_s, N_p] (row stride lda), B_s = X [m_s, K_p] (row stride ldb)
// and optionally db_p[N_p] += column sums of segment 0's A (the bias gradient: only the primal adjoint stream carries it).
// Up to GROUP_MAX_SEG segments: the two adjoint streams of the gradient-penalty pass (mu with the primal activations, lam with the tangent
// ones), or the H horizon steps of the BPTT, are contracted into the same accumulators instead of being separate launches.
//
// MI355X mapping: 64x64 output tile per 256-thread workgroup (4 waves as 2 x 2, one v_mfma_f32_32x32x2_f32 accumulator block each),
// contraction in stages of 32 rows; both operands are contraction-major in memory ([m][n]: n contiguous), so a stage is two [32][64]
// images loaded with 16-B accesses and stored to LDS as they are (row pitch 68 floats: the fragment reads of 32 consecutive n at one m
// and the 16-B staging writes are both conflict-free); register-staged double buffering.  36 KB of LDS: 4 workgroups per CU.  The
// blockIdx -> (problem, tile) map is a prefix table in the kernel arguments (no device-side descriptor memory, no copy): workgroups walk it
// with scalar loads.  Accumulation order inside a tile is fixed (segment 0 first, stages in order): reproducible.
#include "aid_common.h"

namespace aid {

typedef float g_f32x16 __attribute__((ext_vector_type(16)));
typedef float g_f32x4 __attribute__((ext_vector_type(4)));

constexpr int GB = 32, GT = 64, GLD = GT + 4, GIMG = GB * GLD;

// one [32][64] stage of a contraction-major operand: element (m, c) at G[m * ld + c0 + c]; out-of-range rows / columns read as zero
template <bool VEC>
__device__ __forceinline__ void grp_load(const float* __restrict__ G, int ld, int m0, int c0, int M, int Cn, int tid, g_f32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i, m = f >> 4, cq = f & 15;
        const int gm = m0 + m, gc = c0 + 4 * cq;
        g_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gm < M) {
            const float* p = G + (size_t)gm * ld + gc;
            if (VEC) { if (gc < Cn) v = *reinterpret_cast<const g_f32x4*>(p); }
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (gc + j < Cn) v[j] = p[j];
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void grp_store(float* __restrict__ S, int tid, const g_f32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i, m = f >> 4, cq = f & 15;
        *reinterpret_cast<g_f32x4*>(S + m * GLD + 4 * cq) = r[i];
    }
}

template <bool VEC>
__device__ __forceinline__ void grp_tile(const GroupProb& p, int tn, int tk, int sp, float* smem) {
    float* As = smem;                 // [2][32][68]
    float* Bs = smem + 2 * GIMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int n0 = tn * GT, k0 = tk * GT;
    g_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float colsum = 0.f;
    const bool do_db = p.db != nullptr && tk == 0;
    g_f32x4 ra[2], rb[2];
    for (int sg = 0; sg < p.nseg; ++sg) {
        GroupSeg q = p.seg[sg];
        if (p.split > 1) {                                 // this workgroup's chunk of the contraction rows (whole 32-row stages)
            const int chunk = ((q.m + p.split - 1) / p.split + GB - 1) / GB * GB, r0 = min(q.m, sp * chunk), r1 = min(q.m, r0 + chunk);
            q.a += (size_t)r0 * q.lda; q.b += (size_t)r0 * q.ldb; q.m = r1 - r0;
        }
        const int nst = (q.m + GB - 1) / GB;
        if (nst == 0) continue;
        grp_load<VEC>(q.a, q.lda, 0, n0, q.m, p.N, tid, ra);
        grp_load<VEC>(q.b, q.ldb, 0, k0, q.m, p.K, tid, rb);
        __syncthreads();                                   // the previous segment's last stage has been read by every wave
        grp_store(As, tid, ra); grp_store(Bs, tid, rb);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int cur = st & 1;
            if (st + 1 < nst) {
                grp_load<VEC>(q.a, q.lda, (st + 1) * GB, n0, q.m, p.N, tid, ra);
                grp_load<VEC>(q.b, q.ldb, (st + 1) * GB, k0, q.m, p.K, tid, rb);
            }
            const float* a_s = As + cur * GIMG;
            const float* b_s = Bs + cur * GIMG;
            if (do_db && sg == 0 && tid < GT) {            // bias gradient: column sums of dY (fixed order: m ascending)
#pragma unroll 8
                for (int m = 0; m < GB; ++m) colsum += a_s[m * GLD + tid];
            }
#pragma unroll
            for (int g8 = 0; g8 < GB / 8; ++g8) {
                float av[4], bv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    av[j] = a_s[(8 * g8 + 4 * h + j) * GLD + wn + li];
                    bv[j] = b_s[(8 * g8 + 4 * h + j) * GLD + wk + li];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
            }
            if (st + 1 < nst) { grp_store(As + (cur ^ 1) * GIMG, tid, ra); grp_store(Bs + (cur ^ 1) * GIMG, tid, rb); }
            __syncthreads();
        }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma clang loop unroll(full)
    for (int r = 0; r < 16; ++r) {
        const int row = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * h, col = k0 + wk + li;
        if (row < p.N && col < p.K) {
            if (p.split > 1) p.part[((size_t)sp * p.N + row) * p.K + col] = acc[r];
            else { float* c = p.C + (size_t)row * p.ldc + col; *c = *c + acc[r]; }
        }
    }
    if (do_db && tid < GT && n0 + tid < p.N) {
        if (p.split > 1) p.part[(size_t)p.split * p.N * p.K + (size_t)sp * p.N + n0 + tid] = colsum;
        else p.db[n0 + tid] += colsum;
    }
}

__global__ void __launch_bounds__(256) gemm_group_tn_kernel(GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    int pi = 0;
    while (pi + 1 < g.nprob && b >= g.p[pi + 1].tile0) ++pi;          // uniform scalar walk over the prefix table
    const GroupProb& p = g.p[pi];
    const int per = p.tiles_n * p.tiles_k, t = b - p.tile0, sp = t / per, tt = t % per, tn = tt / p.tiles_k, tk = tt % p.tiles_k;
    if (p.vec) grp_tile<true>(p, tn, tk, sp, smem);
    else grp_tile<false>(p, tn, tk, sp, smem);
}

// second pass for split problems: C += part[0] + part[1] + ... (chunk order: reproducible), db likewise
__global__ void __launch_bounds__(256) gemm_group_fold_kernel(GroupArgs g) {
    const GroupProb& p = g.p[blockIdx.y];
    if (p.split <= 1) return;
    const size_t nk = (size_t)p.N * p.K, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += stride) {
        float v = 0.f;
        for (int sp = 0; sp < p.split; ++sp) v += p.part[(size_t)sp * nk + i];
        const size_t row = i / p.K, col = i % p.K;
        float* c = p.C + row * p.ldc + col; *c = *c + v;
    }
    if (p.db && blockIdx.x == 0)
        for (int n = threadIdx.x; n < p.N; n += blockDim.x) {
            float v = 0.f;
            for (int sp = 0; sp < p.split; ++sp) v += p.part[(size_t)p.split * nk + (size_t)sp * p.N + n];
            p.db[n] += v;
        }
}

int GradQueue::add(float* dW, int ldc, int N, int K, const float* dy, int lddy, const float* x, int ldx, int M, float* db) {
    if (M <= 0 || N <= 0 || K <= 0) return AID_OK;
    GroupSeg sgm; sgm.a = dy; sgm.b = x; sgm.lda = lddy; sgm.ldb = ldx; sgm.m = M; sgm._pad = 0;
    const bool al = ((((uintptr_t)dy | (uintptr_t)x) & 15) == 0) && lddy % 4 == 0 && ldx % 4 == 0 && N % 4 == 0 && K % 4 == 0;
    // a further contribution to the gradient the last problem already accumulates (the tangent-adjoint stream, the next horizon step):
    // one more segment of the same tile loop
    if (args.nprob > 0) {
        GroupProb& last = args.p[args.nprob - 1];
        if (last.C == dW && last.N == N && last.K == K && last.ldc == ldc && last.nseg < GROUP_MAX_SEG && !db) {
            last.seg[last.nseg++] = sgm; last.vec = last.vec && al; flop += 2.0 * M * N * K;
            return AID_OK;
        }
    }
    // the same gradient recorded twice with other problems in between: two workgroups of one launch would race on its tiles - close the
    // launch first (stream order then serialises the two accumulations)
    for (int i = 0; i < args.nprob; ++i)
        if (args.p[i].C == dW) { AID_TRY(flush()); break; }
    if (args.nprob == GROUP_MAX_PROB) AID_TRY(flush());
    GroupProb& p = args.p[args.nprob++];
    memset(&p, 0, sizeof(p));
    p.C = dW; p.db = db; p.ldc = ldc; p.N = N; p.K = K; p.nseg = 1; p.seg[0] = sgm; p.vec = al ? 1 : 0;
    p.tiles_k = (K + GT - 1) / GT; p.tiles_n = (N + GT - 1) / GT; p.split = 1; p.part = nullptr;
    flop += 2.0 * M * N * K;
    return AID_OK;
}

int GradQueue::flush() {
    if (args.nprob == 0) return AID_OK;
    // a long contraction over few output tiles (the BPTT's H*K*B = 12 800 rows onto a 512 x 512 gradient: 64 tiles x 400 stages) leaves most CUs
    // idle and the busy ones latency-bound: cut its rows into chunks of >= 1 024, one set of tiles per chunk, folded by a second small launch
    int tiles = 0, nsplit = 0;
    size_t max_nk = 0;
    for (int i = 0; i < args.nprob; ++i) {
        GroupProb& p = args.p[i];
        const int m = p.seg[0].m;
        if (p.nseg == 1 && m >= 4096 && scratch) {
            int sp = m / 1536; sp = sp > 8 ? 8 : sp;
            const size_t need = (size_t)sp * ((size_t)p.N * p.K + p.N);
            if (sp > 1 && scratch_used + need <= scratch_floats) {
                p.split = sp; p.part = scratch + scratch_used; scratch_used += need; ++nsplit;
                if ((size_t)p.N * p.K > max_nk) max_nk = (size_t)p.N * p.K;
            }
        }
        p.tile0 = tiles; tiles += p.tiles_n * p.tiles_k * p.split;
    }
    {
        ProfScope prof(PROF_GEMM_TILE, s, flop, 0.0);
        prof.m = tiles; prof.n = args.nprob; prof.k = 0; prof.layout = 3; prof.variant = 30;
        hipLaunchKernelGGL(gemm_group_tn_kernel, dim3(tiles), dim3(256), sizeof(float) * 4 * GIMG, s, args);
        if (nsplit) {
            const unsigned bx = (unsigned)((max_nk + 1023) / 1024 > 256 ? 256 : (max_nk + 1023) / 1024);
            hipLaunchKernelGGL(gemm_group_fold_kernel, dim3(bx ? bx : 1, args.nprob), dim3(256), 0, s, args);
        }
    }
    args.nprob = 0; flop = 0.0; scratch_used = 0;
    AID_LAUNCH_CHECK();
    return AID_OK;
}

}  // namespace aid

using namespace aid;
extern "C" {
// Test / bench entry for the grouped weight-gradient kernel: `n` problems dW_i[N_i, K_i] += dY_i^T X_i (+ db_i += colsum dY_i), one launch.
int aid_gemm_group_tn(int n, float* const* dW, float* const* db, const float* const* dy, const float* const* x, const int* M, const int* N, const int* K,
                      float* scratch, size_t scratch_floats, void* stream) {
    AID_REQUIRE(n >= 0 && dW && dy && x && M && N && K, AID_E_NULL, "aid_gemm_group_tn: null argument");
    GradQueue q((hipStream_t)stream);
    q.scratch = scratch; q.scratch_floats = scratch ? scratch_floats : 0;
    for (int i = 0; i < n; ++i) AID_TRY(q.add(dW[i], K[i], N[i], K[i], dy[i], N[i], x[i], K[i], M[i], db ? db[i] : nullptr));
    return q.flush();
}
}
