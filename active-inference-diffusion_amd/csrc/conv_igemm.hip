// Implicit-GEMM 3x3 convolution (pad 1, stride 1 or 2) on CNHW activations: forward, data gradient and weight gradient without an im2col
// matrix.  Reference: the four spectral-norm convs of DrQV2Encoder (aid/encoder/visual_encoders.py:135-189) and the DecoderBlock / output
// convs of ConvDecoder (:228-459).
//
//   forward        out[co][n] = sum_k W[co][k] X(k, n)            k = ci 9 + kh 3 + kw,  n = (b, ho, wo)
//                  X(k, n)    = x[ci][b][ho s + kh - 1][wo s + kw - 1]   (0 outside the image)
//   data gradient  of a stride-1 conv = the same kernel on dy with the flipped, channel-transposed weights (conv3_flip_weights)
//   weight grad.   dW[co][k]  = sum_n dy[co][n] X(k, n)            (contraction over n, cut into slabs folded in order: deterministic)
//
// Why: the explicit path wrote X as a [9 Cin][B Ho Wo] matrix (3.8 GB per encoder forward at batch 256, kept for the weight gradient) and read it
// back twice; im2col + col2im were 3.3 ms of a 20 ms forward + backward pair.  Here the B-operand loader computes (b, ho, wo) once per
// thread, keeps a 9-bit border mask, and gathers the 3 x 3 taps straight from the activation planes (231 MB at the widest layer, L2 /
// Infinity-Cache resident across the column tiles that share a plane).
//
// MI355X mapping: 256-thread workgroups, v_mfma_f32_32x32x2_f32, K (or n) in stages of 32, register-staged double buffering with the loads of
// stage t + 1 issued before the MFMAs of stage t.  Loads are unconditional from clamped addresses and masked when written to LDS (a branch
// around a global load makes the compiler drain the memory queue).  Consecutive lanes gather consecutive n (= consecutive wo): coalesced
// 4-B loads for stride 1.
#include "aid_common.h"
#include "mlp.h"

namespace aid {

typedef float ig_f32x16 __attribute__((ext_vector_type(16)));
typedef float ig_f32x4 __attribute__((ext_vector_type(4)));

struct ConvIgArgs {
    const float* W; int ldw;       // [Cout][ldw], ldw >= K = 9 Cin, multiple of 4, zero beyond K
    const float* x;                // [Cin][B][Hi][Wi]
    float* out;                    // [Cout][N]
    int Cin, Cout, B, Hi, Wi, Ho, Wo, stride, K;
    long N;
};

// 4 waves as (4 / WN) x WN; a wave owns TM accumulator blocks of 32 x 32 stacked along the output channels: BM = 32 TM 4 / WN rows, BN = 32 WN columns
// TAPM: the contraction index is tap-major, k' = (kh 3 + kw) Cin + ci, and Cin % BK == 0, so a K stage is BK channels of ONE tap: the tap's offset and
// border validity are computed once per stage and a thread's gathers differ by whole planes only.  (With k = ci 9 + tap every gathered element had its
// own tap: decomposing it, testing its mask bit, selecting its address and carrying (tap, ci) to the next stage was 20 % of the kernel's time on the
// encoder's widest layer - tools/perf/conv_ablation.py: 89 -> 111 TFLOP/s with that arithmetic removed, 113 with the gathers themselves removed.)
// The weights are stored in the same order: conv3_tapmajor_weights / conv3_flip_weights(tap_major).
// VEC4 (TAPM, stride 1, B Ho Wo % 4 == 0): with stride 1 and pad 1 the input index of tap (kh, kw) at output pixel n is n + (kh - 1) W + (kw - 1) - a constant
// offset, rows and images included - so four consecutive pixels of one channel are ONE 16-byte load (4-byte aligned) and one ds_write_b128, masked by the four
// pixels' border bits of that tap: a quarter of the vector-memory and LDS-write instructions of the element-wise gather (the ablation priced the gathers +
// their LDS writes at 12 % of the tap-major kernel).  The two loads in the whole grid that would start before the tensor or end behind it are clamped and
// their elements shifted back when they are written to LDS.
typedef float ig_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int TM, int BK, int WN, int DBG = 0, bool TAPM = false, bool VEC4 = false>
__global__ void __launch_bounds__(256) conv3_igemm_kernel(ConvIgArgs g) {
    constexpr int BM = 32 * TM * (4 / WN), BN = 32 * WN, ALD = BK + 4, BLD = BN + 4, A_FL = BM * ALD, B_FL = BK * BLD;
    constexpr int PA = BM * BK / 4 / 256, KR = 256 / BN, PBG = BK / KR, CPRA = BK / 4;      // float4 A pieces / gathered B elements per thread per stage
    static_assert(PA >= 1 && PA * 1024 == BM * BK && PBG * KR == BK, "tile does not split into whole pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BM][ALD]
    float* Bs = smem + 2 * A_FL;       // [2][BK][BLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave / WN) * (32 * TM), wn = (wave % WN) * 32;
    const long n0 = (long)blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    // ---- this thread's output column: (b, ho, wo), the offset of its top-left tap and the validity of the nine taps
    const int nl = tid % BN, kr = tid / BN;
    const long n = n0 + nl;
    const bool nok = n < g.N;
    const long nn = nok ? n : 0;
    const int wo = (int)(nn % g.Wo); const long t1 = nn / g.Wo; const int ho = (int)(t1 % g.Ho), b = (int)(t1 / g.Ho);
    const int h0 = ho * g.stride - 1, w0 = wo * g.stride - 1;
    const long base = ((long)b * g.Hi + h0) * g.Wi + w0;
    unsigned vm = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            if (nok && h0 + kh >= 0 && h0 + kh < g.Hi && w0 + kw >= 0 && w0 + kw < g.Wi) vm |= 1u << (kh * 3 + kw);
    const long plane = (long)g.B * g.Hi * g.Wi;
    const int nk = (g.K + BK - 1) / BK;

    ig_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    ig_f32x4 ra[PA]; float rb[PBG]; unsigned rbm;
    // tap j of this thread is k = BK kt + kr + KR j: (ci, kh 3 + kw) and the plane offset advance by BK per stage with a carry, no division
    int kr9[PBG]; long kco[PBG];
#pragma unroll
    for (int j = 0; j < PBG; ++j) { const int k = kr + KR * j, ci = k / 9; kr9[j] = k - 9 * ci; kco[j] = (long)ci * plane; }
    int tm_tap = 0, tm_cb = 0;                       // TAPM: the (tap, channel block) of the next stage to load
    const int tm_ncb = TAPM ? g.Cin / BK : 1;
    const long tm_kr = (long)kr * plane, tm_js = (long)KR * plane, tm_cbs = (long)BK * plane;
    // VEC4: this thread's pixel quad (the same for all its pieces) and its k rows kq + KQ i within a stage
    constexpr int QN = BN / 4, KQ = 256 / QN, PQ = VEC4 ? BK / KQ : 1;
    const int vq = tid % QN, vkq = tid / QN;
    ig_f32x4 rbv[PQ]; int rbd[PQ]; unsigned rbn = 0;
    unsigned vlo = 0, vhi = 0;                       // border bits of the quad's four pixels: nibble `tap` of (vhi:vlo) = pixels 0..3 valid at that tap
    const long vn4 = n0 + 4 * vq, vtotal = (long)g.Cin * plane - 4;
    if constexpr (VEC4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long ne = vn4 + e;
            if (ne < g.N) {
                const int wo_ = (int)(ne % g.Wo); const long t_ = ne / g.Wo; const int ho_ = (int)(t_ % g.Ho);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int hh = ho_ + tp / 3 - 1, ww = wo_ + tp % 3 - 1;
                    if (hh >= 0 && hh < g.Hi && ww >= 0 && ww < g.Wi) { if (tp < 8) vlo |= 1u << (4 * tp + e); else vhi |= 1u << e; }
                }
            }
        }
    }
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int f = tid + 256 * i, row = f / CPRA, c4 = f % CPRA;
            const int gr = min(m0 + row, g.Cout - 1), gk = min(kt * BK + 4 * c4, g.ldw - 4);
            if constexpr (DBG != 3) ra[i] = *reinterpret_cast<const ig_f32x4*>(g.W + (size_t)gr * g.ldw + gk);
            else ra[i] = (ig_f32x4){1.f, 1.f, 1.f, 1.f};
        }
        if constexpr (VEC4) {
            const int kh = (tm_tap * 11) >> 5, kw = tm_tap - 3 * kh;
            rbn = (tm_tap < 8 ? vlo >> (4 * tm_tap) : vhi) & 15u;
            const long idx0 = vn4 + (long)(kh - 1) * g.Wi + (kw - 1) + ((long)tm_cb * BK + vkq) * plane;
#pragma unroll
            for (int i = 0; i < PQ; ++i) {
                const long idx = idx0 + (long)(KQ * i) * plane;
                const long idc = idx < 0 ? 0 : (idx > vtotal ? vtotal : idx);
                rbd[i] = (int)(idc - idx);
                rbv[i] = *reinterpret_cast<const ig_f32x4u*>(g.x + idc);
            }
            if (++tm_cb == tm_ncb) { tm_cb = 0; ++tm_tap; }
            return;
        }
        if constexpr (TAPM) {
            const int kh = (tm_tap * 11) >> 5, kw = tm_tap - 3 * kh;
            const bool ok = (vm >> tm_tap) & 1u;
            const float* src = ok ? g.x + (base + kh * g.Wi + kw + tm_kr + (long)tm_cb * tm_cbs) : g.x;
            const long js = ok ? tm_js : 0;
#pragma unroll
            for (int j = 0; j < PBG; ++j) rb[j] = src[(long)j * js];
            rbm = ok ? 0xffffffffu : 0u;
            if (++tm_cb == tm_ncb) { tm_cb = 0; ++tm_tap; }
            return;
        }
        rbm = 0;
#pragma unroll
        for (int j = 0; j < PBG; ++j) {
            const int k = kt * BK + kr + KR * j;
            const int r9 = kr9[j], kh = (r9 * 11) >> 5, kw = r9 - 3 * kh;          // r9 / 3 for r9 < 9
            const bool ok = k < g.K && ((vm >> r9) & 1u);
            if constexpr (DBG == 1 || DBG == 2) rb[j] = 1.f;
            else if constexpr (DBG == 4) rb[j] = g.x[base + (long)j * plane + g.Wi + 1];      // (>= 0: base >= -Wi - 1)
            else rb[j] = g.x[ok ? kco[j] + base + kh * g.Wi + kw : 0];
            rbm |= (ok ? 1u : 0u) << j;
            // k += BK for the next stage (stages are loaded in order, each exactly once)
            kr9[j] += BK % 9; kco[j] += (long)(BK / 9) * plane;
            if (kr9[j] >= 9) { kr9[j] -= 9; kco[j] += plane; }
        }
    };
    auto store_stage = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int f = tid + 256 * i, row = f / CPRA, c4 = f % CPRA;
            ig_f32x4 v = ra[i];
            if (m0 + row >= g.Cout || kt * BK + 4 * c4 >= g.ldw) v = (ig_f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<ig_f32x4*>(As + buf * A_FL + row * ALD + 4 * c4) = v;
        }
        if constexpr (VEC4) {
#pragma unroll
            for (int i = 0; i < PQ; ++i) {
                ig_f32x4 v = rbv[i];
                const int d = rbd[i];
                if (d != 0) {                        // a load clamped at an end of the tensor (two in the grid): element e wants loaded[e - d]
                    ig_f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const int j = e - d; w[e] = j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : (j == 3 ? v[3] : 0.f))); }
                    v = w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((rbn >> e) & 1u) ? v[e] : 0.f;
                *reinterpret_cast<ig_f32x4*>(Bs + buf * B_FL + (vkq + KQ * i) * BLD + 4 * vq) = v;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < PBG; ++j) { if constexpr (DBG != 2) Bs[buf * B_FL + (kr + KR * j) * BLD + nl] = ((rbm >> j) & 1u) ? rb[j] : 0.f; }
    };
    load_stage(0); store_stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_stage(kt + 1);
        const float* a_s = As + cur * A_FL;
        const float* b_s = Bs + cur * B_FL;
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
            ig_f32x4 av[TM]; float bv[4];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const ig_f32x4*>(a_s + (wm + 32 * i + li) * ALD + 8 * g8 + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = b_s[(8 * g8 + 4 * h + j) * BLD + wn + li];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[q], acc[i], 0, 0, 0);
        }
        if (kt + 1 < nk) store_stage(kt + 1, cur ^ 1);
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const long col = n0 + wn + li;
#pragma clang loop unroll(full)
    for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < g.Cout && col < g.N) g.out[(size_t)row * g.N + col] = acc[i][r];
        }
}

static int g_ig_dbg = 0;
static int g_dw_tm = 1;          // aid_conv_debug(12 / 13): tap-major weight-gradient kernel off / on (A/B timing)
static int g_dw_big = 1;         // aid_conv_debug(14 / 15 / 16): its 128 x 128-tile form off / on where 9 Cin % 128 == 0 / on wherever Cin % 64 == 0 (A/B timing, tests)
static int g_ig_vec4 = 1;        // aid_conv_debug(10 / 11): four-pixel vector gathers of the tap-major kernels off / on (A/B timing)
// tap-major kernels need whole BK-channel blocks per tap: BK = 32 for the <= 32-output-channel tile, 16 otherwise
bool conv3_tapmajor_ok(int Cin, int Cout) { return Cin % (Cout <= 32 ? 32 : 16) == 0; }
int conv3_igemm_ex(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, int w_tap_major, hipStream_t s);
int conv3_igemm(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, hipStream_t s) {
    return conv3_igemm_ex(W, ldw, x, out, Cin, Cout, B, Hi, Ho, stride, 0, s);
}
// w_tap_major: W is [Cout][(kh 3 + kw) Cin + ci] (conv3_tapmajor_weights); needs conv3_tapmajor_ok(Cin, Cout)
int conv3_igemm_ex(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, int w_tap_major, hipStream_t s) {
    AID_REQUIRE(W && x && out, AID_E_NULL, "conv3_igemm: null operand");
    AID_REQUIRE(Cin > 0 && Cout > 0 && B > 0 && (stride == 1 || stride == 2) && ldw >= 9 * Cin && ldw % 4 == 0 && ((((uintptr_t)W) & 15) == 0), AID_E_SHAPE,
                "conv3_igemm: Cin=%d Cout=%d stride=%d ldw=%d", Cin, Cout, stride, ldw);
    AID_REQUIRE(!w_tap_major || conv3_tapmajor_ok(Cin, Cout), AID_E_SHAPE, "conv3_igemm: tap-major weights need Cin %% %d == 0 (Cin=%d)", Cout <= 32 ? 32 : 16, Cin);
    ConvIgArgs g; memset(&g, 0, sizeof(g));
    g.W = W; g.ldw = ldw; g.x = x; g.out = out; g.Cin = Cin; g.Cout = Cout; g.B = B; g.Hi = Hi; g.Wi = Hi; g.Ho = Ho; g.Wo = Ho; g.stride = stride; g.K = 9 * Cin;
    g.N = (long)B * Ho * Ho;
    ProfScope prof(PROF_GEMM_TILE, s, 2.0 * Cout * g.K * (double)g.N, 4.0 * ((double)Cout * g.K + (double)Cin * B * Hi * Hi + (double)Cout * g.N));
    prof.m = Cout; prof.n = (int)(g.N > 2000000000L ? 2000000000L : g.N); prof.k = g.K; prof.layout = stride + (w_tap_major ? 8 : 0); prof.variant = 31;
    // 128 x 64 tiles in 16-deep stages (29 KB of LDS: five workgroups per CU; measured 93 TFLOP/s on the 256 x 451 584 x 1152 layer against 83
    // with 32-deep stages at two per CU and 74 with 64 x 64 tiles, which gather every column four times); 64-row tiles for Cout <= 64
    // up to 32 output channels: 32 x 128 tiles (a 64-row tile would be half empty); up to 64: 64 x 64 in 16-deep stages (19 KB)
#define IG_LAUNCH(TM_, BK_, WN_) do { constexpr int BM_ = 32 * TM_ * (4 / WN_), BN_ = 32 * WN_; constexpr size_t lds = sizeof(float) * 2 * (BM_ * (BK_ + 4) + BK_ * (BN_ + 4)); \
        hipLaunchKernelGGL((conv3_igemm_kernel<TM_, BK_, WN_>), dim3((unsigned)((g.N + BN_ - 1) / BN_), (Cout + BM_ - 1) / BM_), dim3(256), lds, s, g); } while (0)
    if (g_ig_dbg && g_ig_dbg < 10 && Cout > 64) {      // timing experiments (aid_conv_debug; results are wrong by construction)
#define IG_LAUNCH_D(D_) do { constexpr size_t lds = sizeof(float) * 2 * (128 * 20 + 16 * 68); \
        hipLaunchKernelGGL((conv3_igemm_kernel<2, 16, 2, D_>), dim3((unsigned)((g.N + 63) / 64), (Cout + 127) / 128), dim3(256), lds, s, g); } while (0)
        if (g_ig_dbg == 1) IG_LAUNCH_D(1); else if (g_ig_dbg == 2) IG_LAUNCH_D(2); else if (g_ig_dbg == 3) IG_LAUNCH_D(3); else IG_LAUNCH_D(4);
#undef IG_LAUNCH_D
        AID_LAUNCH_CHECK();
        return AID_OK;
    }
#define IG_LAUNCH_T(TM_, BK_, WN_) do { constexpr int BM_ = 32 * TM_ * (4 / WN_), BN_ = 32 * WN_; constexpr size_t lds = sizeof(float) * 2 * (BM_ * (BK_ + 4) + BK_ * (BN_ + 4)); \
        hipLaunchKernelGGL((conv3_igemm_kernel<TM_, BK_, WN_, 0, true>), dim3((unsigned)((g.N + BN_ - 1) / BN_), (Cout + BM_ - 1) / BM_), dim3(256), lds, s, g); } while (0)
#define IG_LAUNCH_V(TM_, BK_, WN_) do { constexpr int BM_ = 32 * TM_ * (4 / WN_), BN_ = 32 * WN_; constexpr size_t lds = sizeof(float) * 2 * (BM_ * (BK_ + 4) + BK_ * (BN_ + 4)); \
        hipLaunchKernelGGL((conv3_igemm_kernel<TM_, BK_, WN_, 0, true, true>), dim3((unsigned)((g.N + BN_ - 1) / BN_), (Cout + BM_ - 1) / BM_), dim3(256), lds, s, g); } while (0)
    const bool vec4 = w_tap_major && stride == 1 && Hi == Ho && g.N % 4 == 0 && (long)Cin * B * Hi * Hi >= 8 && g_ig_vec4;
    if (vec4) {
        if (Cout <= 32) IG_LAUNCH_V(1, 32, 4);
        else if (Cout <= 64) IG_LAUNCH_V(1, 16, 2);
        // >= 128 output channels over >= 65 536 pixels: 128 x 128 tiles, a wave owns 128 x 32 (four accumulator blocks per gathered fragment instead of two; 37 KB of
        // LDS): encoder layer 3 forward 105 -> 116 TFLOP/s, its data gradient 112 -> 118, ConvDecoder 128 -> 256 at 84 x 84 104 -> 112; on the 21 x 21 / 42 x 42 planes of
        // 16 images (441 - 1 764 tiles) the smaller tile fills the chip better (82 against 79); 32-deep stages lose everywhere but there (tools/perf/conv_tiles.py)
        else if (Cout >= 128 && g.N >= 65536 && g_ig_dbg != 22) IG_LAUNCH_V(4, 16, 4);
        else IG_LAUNCH_V(2, 16, 2);
    }
    else if (w_tap_major) {
        if (Cout <= 32) IG_LAUNCH_T(1, 32, 4);
        else if (Cout <= 64) IG_LAUNCH_T(1, 16, 2);
        else IG_LAUNCH_T(2, 16, 2);
    }
    else if (Cout <= 32) IG_LAUNCH(1, 32, 4);
    else if (Cout <= 64) IG_LAUNCH(1, 16, 2);
    else IG_LAUNCH(2, 16, 2);
#undef IG_LAUNCH_V
#undef IG_LAUNCH_T
#undef IG_LAUNCH
    AID_LAUNCH_CHECK();
    return AID_OK;
}

// Wt[ci][co 9 + kh 3 + kw] = W[co][ci 9 + (2 - kh) 3 + (2 - kw)], rows zero-padded to ldwt
__global__ void conv3_flip_weights_kernel(const float* __restrict__ W, int ldw, float* __restrict__ Wt, int ldwt, int Cin, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * ldwt) return;
    const int ci = i / ldwt, k = i % ldwt;
    float v = 0.f;
    if (k < Cout * 9) { const int co = k / 9, r9 = k % 9, kh = r9 / 3, kw = r9 % 3; v = W[(size_t)co * ldw + ci * 9 + (2 - kh) * 3 + (2 - kw)]; }
    Wt[i] = v;
}
// tap-major forms.  Wtm[co][tap Cin + ci] = W[co][ci 9 + tap];   flipped: Wt[ci][tap Cout + co] = W[co][ci 9 + (8 - tap)]
__global__ void conv3_tapmajor_weights_kernel(const float* __restrict__ W, int ldw, float* __restrict__ Wtm, int ldwt, int Cin, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * ldwt) return;
    const int co = i / ldwt, k = i % ldwt;
    float v = 0.f;
    if (k < Cin * 9) { const int tap = k / Cin, ci = k % Cin; v = W[(size_t)co * ldw + ci * 9 + tap]; }
    Wtm[i] = v;
}
__global__ void conv3_flip_weights_tm_kernel(const float* __restrict__ W, int ldw, float* __restrict__ Wt, int ldwt, int Cin, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * ldwt) return;
    const int ci = i / ldwt, k = i % ldwt;
    float v = 0.f;
    if (k < Cout * 9) { const int tap = k / Cout, co = k % Cout; v = W[(size_t)co * ldw + ci * 9 + (8 - tap)]; }
    Wt[i] = v;
}
int conv3_tapmajor_weights(const float* W, int ldw, float* Wtm, int ldwt, int Cin, int Cout, hipStream_t s) {
    AID_REQUIRE(ldwt >= 9 * Cin && ldwt % 4 == 0, AID_E_SHAPE, "conv3_tapmajor_weights: ldwt=%d", ldwt);
    hipLaunchKernelGGL(conv3_tapmajor_weights_kernel, dim3((Cout * ldwt + 255) / 256), dim3(256), 0, s, W, ldw, Wtm, ldwt, Cin, Cout);
    AID_LAUNCH_CHECK(); return AID_OK;
}
int conv3_flip_weights(const float* W, int ldw, float* Wt, int ldwt, int Cin, int Cout, hipStream_t s, int tap_major) {
    AID_REQUIRE(ldwt >= 9 * Cout && ldwt % 4 == 0, AID_E_SHAPE, "conv3_flip_weights: ldwt=%d", ldwt);
    if (tap_major) hipLaunchKernelGGL(conv3_flip_weights_tm_kernel, dim3((Cin * ldwt + 255) / 256), dim3(256), 0, s, W, ldw, Wt, ldwt, Cin, Cout);
    else hipLaunchKernelGGL(conv3_flip_weights_kernel, dim3((Cin * ldwt + 255) / 256), dim3(256), 0, s, W, ldw, Wt, ldwt, Cin, Cout);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// ---- weight gradient: slab[sy][co][k] = sum over this slab's n of dy[co][n] X(k, n); 64 x 64 output tile, n in stages of 32 ------------------
struct ConvDwArgs {
    const float* dy;               // [Cout][N]
    const float* x;                // [Cin][B][Hi][Wi]
    float* slab; int ldw;          // [S][Cout][ldw]
    int Cin, Cout, B, Hi, Wi, Ho, Wo, stride, K, tiles_k;
    long N, chunk;                 // n per slab (multiple of 32)
};
__global__ void __launch_bounds__(256) conv3_igemm_dw_kernel(ConvDwArgs g) {
    constexpr int BT = 64, BK = 32, LD = BK + 4, FL = BT * LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;              // [2][64 co][36]   (n contiguous)
    float* Bs = smem + 2 * FL;     // [2][64 k][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int tco = blockIdx.x / g.tiles_k, tk = blockIdx.x % g.tiles_k;
    const int co0 = tco * BT, k0 = tk * BT;
    const long nb0 = (long)blockIdx.y * g.chunk, nb1 = min(g.N, nb0 + g.chunk);
    const int nst = (int)((nb1 - nb0 + BK - 1) / BK);
    const long plane = (long)g.B * g.Hi * g.Wi;
    const int nl = tid & 31, kr = tid >> 5;
    // the 8 taps this thread gathers are fixed for the whole kernel: k = k0 + kr + 8 j
    long toff[8]; int tsh[8];                                   // offset of the tap relative to (b, ho s, wo s); bit positions of its row / column validity
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + kr + 8 * j, kc = min(k, g.K - 1), ci = kc / 9, r9 = kc - 9 * ci, dh = r9 / 3, dw = r9 % 3;
        toff[j] = (long)ci * plane + (long)(dh - 1) * g.Wi + (dw - 1);
        tsh[j] = k < g.K ? dh + 4 * dw : 31;                    // 31: an always-clear bit (k beyond the layer's K)
    }
    ig_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ig_f32x4 ra[2]; float rb[8]; unsigned rbm; bool ram[2];
    // this thread's column n = nb0 + 32 st + nl as (b, ho, wo), advanced by 32 per stage with carries (no division in the loop)
    int cwo, cho, cb;
    { const long nn = min(nb0 + nl, g.N - 1); cwo = (int)(nn % g.Wo); const long t1 = nn / g.Wo; cho = (int)(t1 % g.Ho); cb = (int)(t1 / g.Ho); }
    auto load_stage = [&](int st) {
        const long nb = nb0 + (long)st * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            const long nn = nb + 4 * c4;
            ram[i] = co0 + row < g.Cout && nn + 3 < nb1;            // N and the slab bounds are multiples of 4 on the encoder / decoder shapes
            const long a = (long)min(co0 + row, g.Cout - 1) * g.N + (ram[i] ? nn : 0);
            ra[i] = *reinterpret_cast<const ig_f32x4*>(g.dy + a);
        }
        const long n = nb + nl;
        const bool nok = n < nb1;
        const int wo = cwo, ho = cho, b = min(cb, g.B - 1);
        cwo += BK;                                          // next stage's column
        while (cwo >= g.Wo) { cwo -= g.Wo; if (++cho == g.Ho) { cho = 0; ++cb; } }
        const int hc = ho * g.stride, wc = wo * g.stride;
        const long base = ((long)b * g.Hi + hc) * g.Wi + wc;
        // validity of the three rows / columns around (hc, wc): bits 0..2 rows, as a 3 x 3 product table in bits dh + 4 dw
        unsigned hb = 0, vb = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) { if (hc + d - 1 >= 0 && hc + d - 1 < g.Hi) hb |= 1u << d; }
#pragma unroll
        for (int d = 0; d < 3; ++d) { if (nok && wc + d - 1 >= 0 && wc + d - 1 < g.Wi) vb |= hb << (4 * d); }
        rbm = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = (vb >> tsh[j]) & 1u;
            rb[j] = g.x[ok ? base + toff[j] : 0];
            rbm |= (ok ? 1u : 0u) << j;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            *reinterpret_cast<ig_f32x4*>(As + buf * FL + row * LD + 4 * c4) = ram[i] ? ra[i] : (ig_f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[buf * FL + (kr + 8 * j) * LD + nl] = ((rbm >> j) & 1u) ? rb[j] : 0.f;
    };
    if (nst > 0) { load_stage(0); store_stage(0); }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) load_stage(st + 1);
        const float* a_s = As + cur * FL;
        const float* b_s = Bs + cur * FL;
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
            const ig_f32x4 av = *reinterpret_cast<const ig_f32x4*>(a_s + (wm + li) * LD + 8 * g8 + 4 * h);
            const ig_f32x4 bv = *reinterpret_cast<const ig_f32x4*>(b_s + (wk + li) * LD + 8 * g8 + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
        if (st + 1 < nst) store_stage(cur ^ 1);
        __syncthreads();
    }
    float* out = g.slab + (size_t)blockIdx.y * g.Cout * g.ldw;
#pragma clang loop unroll(full)
    for (int r = 0; r < 16; ++r) {
        const int row = co0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h, col = k0 + wk + li;
        if (row < g.Cout && col < g.ldw) out[(size_t)row * g.ldw + col] = col < g.K ? acc[r] : 0.f;
    }
}
// The same contraction for stride-1 layers with Cin % 64 == 0, tap-major: the 64 columns of a workgroup's output tile are 64 CHANNELS of ONE tap, k' = tap Cin + ci,
// so the gathered operand of a stage is 64 channel rows of 32 consecutive pixels shifted by the tap's constant offset (stride 1: input index = n + (dh - 1) W + (dw - 1)):
// two 4-byte-aligned 16-byte loads per thread and two ds_write_b128 instead of eight element gathers with their own taps, and ONE border test per pixel for the
// workgroup's tap.  The slabs come out in k' order; conv3_dw_fold_kernel puts them back into the weight's [co][ci 9 + tap] layout while it folds them.
__global__ void __launch_bounds__(256) conv3_igemm_dw_tm_kernel(ConvDwArgs g) {
    constexpr int BT = 64, BK = 32, LD = BK + 4, FL = BT * LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;              // [2][64 co][36]   (n contiguous)
    float* Bs = smem + 2 * FL;     // [2][64 k'][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int tco = blockIdx.x / g.tiles_k, tk = blockIdx.x % g.tiles_k;
    const int co0 = tco * BT, k0 = tk * BT;                     // k0 = tap Cin + ci0 (Cin % 64 == 0: one tap per tile)
    const int tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin, dh = tap / 3, dw = tap - 3 * dh;
    const long nb0 = (long)blockIdx.y * g.chunk, nb1 = min(g.N, nb0 + g.chunk);
    const int nst = (int)((nb1 - nb0 + BK - 1) / BK);
    const long plane = (long)g.B * g.Hi * g.Wi, total4 = (long)g.Cin * plane - 4;
    const int q = tid & 7, krow = tid >> 3;                     // this thread's pixel quad within a stage and its two channel rows krow, krow + 32
    const long toff = (long)(dh - 1) * g.Wi + (dw - 1) + (long)(ci0 + krow) * plane;
    ig_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ig_f32x4 ra[2], rbv[2]; int rbd[2]; unsigned rbn = 0; bool ram[2];
    // (wo, ho) of the quad's first pixel n = nb0 + 32 st + 4 q, advanced by 32 per stage with carries
    int cwo, cho;
    { const long nn = min(nb0 + 4 * q, g.N - 1); cwo = (int)(nn % g.Wo); cho = (int)((nn / g.Wo) % g.Ho); }
    auto load_stage = [&](int st) {
        const long nb = nb0 + (long)st * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            const long nn = nb + 4 * c4;
            ram[i] = co0 + row < g.Cout && nn + 3 < nb1;
            const long a = (long)min(co0 + row, g.Cout - 1) * g.N + (ram[i] ? nn : 0);
            ra[i] = *reinterpret_cast<const ig_f32x4*>(g.dy + a);
        }
        const long n4 = nb + 4 * q;
        // border bits of the quad's four pixels at this workgroup's tap
        rbn = 0;
        {
            int wo = cwo, ho = cho;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int hh = ho + dh - 1, ww = wo + dw - 1;
                if (n4 + e < nb1 && hh >= 0 && hh < g.Hi && ww >= 0 && ww < g.Wi) rbn |= 1u << e;
                if (++wo == g.Wo) { wo = 0; if (++ho == g.Ho) ho = 0; }
            }
        }
        cwo += BK;                                              // next stage's quad
        while (cwo >= g.Wo) { cwo -= g.Wo; if (++cho == g.Ho) cho = 0; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long idx = toff + n4 + (long)(32 * i) * plane;
            const long idc = idx < 0 ? 0 : (idx > total4 ? total4 : idx);
            rbd[i] = (int)(idc - idx);
            rbv[i] = *reinterpret_cast<const ig_f32x4u*>(g.x + idc);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            *reinterpret_cast<ig_f32x4*>(As + buf * FL + row * LD + 4 * c4) = ram[i] ? ra[i] : (ig_f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ig_f32x4 v = rbv[i];
            const int d = rbd[i];
            if (d != 0) {                                        // a load clamped at an end of the tensor: element e wants loaded[e - d]
                ig_f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int j = e - d; w[e] = j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : (j == 3 ? v[3] : 0.f))); }
                v = w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((rbn >> e) & 1u) ? v[e] : 0.f;
            *reinterpret_cast<ig_f32x4*>(Bs + buf * FL + (krow + 32 * i) * LD + 4 * q) = v;
        }
    };
    if (nst > 0) { load_stage(0); store_stage(0); }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) load_stage(st + 1);
        const float* a_s = As + cur * FL;
        const float* b_s = Bs + cur * FL;
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
            const ig_f32x4 av = *reinterpret_cast<const ig_f32x4*>(a_s + (wm + li) * LD + 8 * g8 + 4 * h);
            const ig_f32x4 bv = *reinterpret_cast<const ig_f32x4*>(b_s + (wk + li) * LD + 8 * g8 + 4 * h);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[qq], bv[qq], acc, 0, 0, 0);
        }
        if (st + 1 < nst) store_stage(cur ^ 1);
        __syncthreads();
    }
    float* out = g.slab + (size_t)blockIdx.y * g.Cout * g.ldw;
#pragma clang loop unroll(full)
    for (int r = 0; r < 16; ++r) {
        const int row = co0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h, col = k0 + wk + li;
        if (row < g.Cout && col < g.ldw) out[(size_t)row * g.ldw + col] = col < g.K ? acc[r] : 0.f;
    }
}
// (round 5) The tap-major weight gradient on 128 x 128 output tiles: 128 output channels x 128 positions k' = tap Cin + ci (Cin % 64 == 0: each 64-row half of
// the tile lies in ONE tap), n in stages of 32 with PREFETCH DISTANCE 2 (stage st + 2's global loads are issued before stage st is contracted and land in the
// register set stage st + 1 does not use) and 4 accumulators of 32 x 32 per wave - per workgroup barrier 64 MFMAs per wave against 16 on the 64 x 64 tile, whose
// barrier convoy left the matrix pipe idle a quarter of the time (81 TFLOP/s on the encoder's 128 -> 256 layer, 3.26 ms per pass).  Operand traffic per FLOP
// halves (a 128 x 128 tile reads 2 x 128 rows per 32 n for 4x the products of a 64 x 64 tile).  74 KB of LDS, 2 workgroups per CU; the slab count is chosen
// so that the grid is one or two full rounds of 512 workgroups.  Same slabs, same fold (conv3_dw_fold_tm_kernel), same summation order within a slab as the
// 64 x 64 kernel: n ascending, two k-values per MFMA.
__global__ void __launch_bounds__(256) conv3_igemm_dw_big_kernel(ConvDwArgs g) {
    constexpr int BT = 128, BK = 32, LD = BK + 4, FL = BT * LD, P = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;              // [2][128 co][36]   (n contiguous)
    float* Bs = smem + 2 * FL;     // [2][128 k'][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int tco = blockIdx.x / g.tiles_k, tk = blockIdx.x % g.tiles_k;
    const int co0 = tco * BT, k0 = tk * BT;
    const long nb0 = (long)blockIdx.y * g.chunk, nb1 = min(g.N, nb0 + g.chunk);
    const int nst = (int)((nb1 - nb0 + BK - 1) / BK);
    const long plane = (long)g.B * g.Hi * g.Wi, total4 = (long)g.Cin * plane - 4;
    const int q = tid & 7, krow = tid >> 3;                     // this thread's pixel quad within a stage; its rows krow + 32 i of either operand
    // per piece i: rows k0 + krow + 32 i -> (tap, ci); pieces 0, 1 share the tile's first tap, pieces 2, 3 the tap of its second half
    long toff[P]; int dh_[2], dw_[2]; bool kok[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = k0 + krow + 32 * i;
        kok[i] = k < g.K;
        const int kc = kok[i] ? k : 0, tap = kc / g.Cin, ci = kc - tap * g.Cin, dh = tap / 3, dw = tap - 3 * dh;
        toff[i] = (long)(dh - 1) * g.Wi + (dw - 1) + (long)ci * plane;
        if ((i & 1) == 0) { dh_[i >> 1] = dh; dw_[i >> 1] = dw; }
    }
    const bool one_tap = dh_[0] == dh_[1] && dw_[0] == dw_[1];
    ig_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Stage { ig_f32x4 a[P], b[P]; int d[P]; unsigned am, bm0, bm1; };
    int cwo, cho;
    { const long nn = min(nb0 + 4 * q, g.N - 1); cwo = (int)(nn % g.Wo); cho = (int)((nn / g.Wo) % g.Ho); }
    auto border = [&](int wo, int ho, int dh, int dw, bool inr) {
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int hh = ho + dh - 1, ww = wo + dw - 1;
            if (inr && hh >= 0 && hh < g.Hi && ww >= 0 && ww < g.Wi) m |= 1u << e;
            if (++wo == g.Wo) { wo = 0; if (++ho == g.Ho) ho = 0; }
        }
        return m;
    };
    // the loads of stage st (stages are requested in order, so one (wo, ho) tracker serves them all); unconditional, from clamped addresses
    auto load_stage = [&](Stage& t, int st) {
        const long nb = nb0 + (long)st * BK, n4 = nb + 4 * q;
        const bool inr = n4 < nb1;                              // N and the slab bounds are multiples of 4: a quad is inside or outside as a whole
        t.am = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int row = krow + 32 * i;
            const bool ok = co0 + row < g.Cout && inr;
            t.am |= (ok ? 1u : 0u) << i;
            t.a[i] = *reinterpret_cast<const ig_f32x4*>(g.dy + (long)min(co0 + row, g.Cout - 1) * g.N + (ok ? n4 : 0));
        }
        t.bm0 = border(cwo, cho, dh_[0], dw_[0], inr);
        t.bm1 = one_tap ? t.bm0 : border(cwo, cho, dh_[1], dw_[1], inr);
        cwo += BK;
        while (cwo >= g.Wo) { cwo -= g.Wo; if (++cho == g.Ho) cho = 0; }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const long idx = toff[i] + n4;
            const long idc = idx < 0 ? 0 : (idx > total4 ? total4 : idx);
            t.d[i] = (int)(idc - idx);
            t.b[i] = *reinterpret_cast<const ig_f32x4u*>(g.x + idc);
        }
    };
    auto store_stage = [&](const Stage& t, int buf) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            *reinterpret_cast<ig_f32x4*>(As + buf * FL + (krow + 32 * i) * LD + 4 * q) = ((t.am >> i) & 1u) ? t.a[i] : (ig_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < P; ++i) {
            ig_f32x4 v = t.b[i];
            const int d = t.d[i];
            if (d != 0) {                                        // a load clamped at an end of the tensor: element e wants loaded[e - d]
                ig_f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int j = e - d; w[e] = j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : (j == 3 ? v[3] : 0.f))); }
                v = w;
            }
            const unsigned m = kok[i] ? (i < 2 ? t.bm0 : t.bm1) : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((m >> e) & 1u) ? v[e] : 0.f;
            *reinterpret_cast<ig_f32x4*>(Bs + buf * FL + (krow + 32 * i) * LD + 4 * q) = v;
        }
    };
    auto contract = [&](int cur) {
        const float* a_s = As + cur * FL;
        const float* b_s = Bs + cur * FL;
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
            ig_f32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const ig_f32x4*>(a_s + (wm + 32 * i + li) * LD + 8 * g8 + 4 * h);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const ig_f32x4*>(b_s + (wk + 32 * j + li) * LD + 8 * g8 + 4 * h);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][qq], bv[j][qq], acc[i][j], 0, 0, 0);
        }
    };
    Stage s0, s1;
    load_stage(s0, 0);
    load_stage(s1, 1);                                          // stages beyond the slab come back as zeros (their quads are outside [nb0, nb1))
    store_stage(s0, 0);
    __syncthreads();
    // an odd stage count runs one extra all-zero stage instead of leaving the loop half way (the accumulators would otherwise shuttle between register banks)
    const int nst2 = (nst + 1) & ~1;
    for (int st = 0; st < nst2; st += 2) {
        load_stage(s0, st + 2);
        __builtin_amdgcn_sched_barrier(0);                      // keep the prefetch ahead of the MFMA block
        contract(0);
        store_stage(s1, 1);
        __syncthreads();
        load_stage(s1, st + 3);
        __builtin_amdgcn_sched_barrier(0);
        contract(1);
        store_stage(s0, 0);
        __syncthreads();
    }
    float* out = g.slab + (size_t)blockIdx.y * g.Cout * g.ldw;
#pragma clang loop unroll(full)
    for (int i = 0; i < 2; ++i)
#pragma clang loop unroll(full)
        for (int j = 0; j < 2; ++j)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, col = k0 + wk + 32 * j + li;
                if (row < g.Cout && col < g.ldw) out[(size_t)row * g.ldw + col] = col < g.K ? acc[i][j][r] : 0.f;
            }
}
// dW[co][k] = sum of the S slabs (slab order); tm_cin > 0: the slabs are tap-major (position tap tm_cin + ci), dW gets the weight's own order k = ci 9 + tap
__global__ void conv3_dw_fold_tm_kernel(const float* __restrict__ slab, int S, int Cout, int ldw, int tm_cin, float* __restrict__ dW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)Cout * ldw;
    if (i >= n) return;
    const int co = (int)(i / ldw), k = (int)(i % ldw);
    float v = 0.f;
    if (k < 9 * tm_cin) {
        const int ci = k / 9, tap = k - 9 * ci;
        const size_t src = (size_t)co * ldw + tap * tm_cin + ci;
        for (int sy = 0; sy < S; ++sy) v += slab[(size_t)sy * n + src];
    }
    dW[i] = v;
}
__global__ void conv3_dw_fold_kernel(const float* __restrict__ slab, int S, size_t n, float* __restrict__ dW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int sy = 0; sy < S; ++sy) v += slab[(size_t)sy * n + i];
    dW[i] = v;
}
int conv3_igemm_dw(const float* dy, const float* x, float* dW, int ldw, int Cin, int Cout, int B, int Hi, int Ho, int stride, float* slab, int S, hipStream_t s) {
    AID_REQUIRE(dy && x && dW && slab && S >= 1, AID_E_NULL, "conv3_igemm_dw: null operand");
    const long N = (long)B * Ho * Ho;
    AID_REQUIRE(N % 4 == 0 && ldw % 4 == 0 && ldw >= 9 * Cin && ((((uintptr_t)dy) & 15) == 0), AID_E_SHAPE, "conv3_igemm_dw: N=%ld ldw=%d", N, ldw);
    ConvDwArgs g; memset(&g, 0, sizeof(g));
    g.dy = dy; g.x = x; g.slab = slab; g.ldw = ldw; g.Cin = Cin; g.Cout = Cout; g.B = B; g.Hi = Hi; g.Wi = Hi; g.Ho = Ho; g.Wo = Ho; g.stride = stride; g.K = 9 * Cin;
    g.N = N; g.tiles_k = (ldw + 63) / 64;
    g.chunk = ((N + S - 1) / S + 31) / 32 * 32;
    AID_REQUIRE(g.chunk >= 32, AID_E_SHAPE, "conv3_igemm_dw: %d slabs for %ld columns", S, N);
    const int S_eff = (int)((N + g.chunk - 1) / g.chunk);
    ProfScope prof(PROF_GEMM_TILE, s, 2.0 * Cout * g.K * (double)N, 4.0 * ((double)Cout * N + (double)Cin * B * Hi * Hi));
    prof.m = Cout; prof.n = g.K; prof.k = (int)(N > 2000000000L ? 2000000000L : N); prof.layout = 4 + stride; prof.variant = 32;
    const size_t n = (size_t)Cout * ldw;
    // 128 x 128 tap-major tiles, prefetch distance 2: measured on the encoder's layers (tools/perf/conv_dw_ab.py, isolated) 128 -> 256: 2.63 ms (101 TFLOP/s)
    // against 2.83 (94) with the 64 x 64 tiles; 64 -> 128 (4.5 tiles of k': the fifth is half empty): 0.81 against 0.79 - so only where 9 Cin is a
    // multiple of 128.  (The kernel itself handles Cin % 64 == 0: tests/test_hip_kernels.py runs it there through aid_conv_debug(16).)
    if (stride == 1 && Hi == Ho && Cin % 64 == 0 && Cout % 128 == 0 && g_dw_tm && g_dw_big && (Cin % 128 == 0 || g_dw_big == 2)) {
        prof.layout = 4 + stride + 16;
        const int tiles = (Cout / 128) * ((ldw + 127) / 128);
        // slabs so that tiles x slabs is two full rounds of the chip's 512 resident workgroups (2 per CU), or - with fewer slabs on offer - nearly two
        // (> 1.5) or exactly one: a round that is a small fraction full costs as much as a full one
        const int r2 = 1024 / tiles, r1 = 512 / tiles;
        int S_use = S_eff >= r2 ? r2 : (S_eff >= r1 ? (S_eff * tiles > 768 ? S_eff : r1) : S_eff);
        if (S_use < 1) S_use = 1;
        g.tiles_k = (ldw + 127) / 128;
        g.chunk = ((N + S_use - 1) / S_use + 31) / 32 * 32;
        const int S_big = (int)((N + g.chunk - 1) / g.chunk);
        const size_t lds = sizeof(float) * 4 * 128 * (32 + 4);
        static bool attr_done = false;
        if (!attr_done) { AID_CHECK_HIP(hipFuncSetAttribute((const void*)conv3_igemm_dw_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_done = true; }
        hipLaunchKernelGGL(conv3_igemm_dw_big_kernel, dim3(tiles, S_big), dim3(256), lds, s, g);
        hipLaunchKernelGGL(conv3_dw_fold_tm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, S_big, Cout, ldw, Cin, dW);
        AID_LAUNCH_CHECK();
        return AID_OK;
    }
    if (stride == 1 && Hi == Ho && Cin % 64 == 0 && g_dw_tm) {      // tap-major tiles + four-pixel vector gathers
        prof.layout = 4 + stride + 8;
        hipLaunchKernelGGL(conv3_igemm_dw_tm_kernel, dim3(((Cout + 63) / 64) * g.tiles_k, S_eff), dim3(256), sizeof(float) * 4 * 64 * (32 + 4), s, g);
        hipLaunchKernelGGL(conv3_dw_fold_tm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, S_eff, Cout, ldw, Cin, dW);
        AID_LAUNCH_CHECK();
        return AID_OK;
    }
    hipLaunchKernelGGL(conv3_igemm_dw_kernel, dim3(((Cout + 63) / 64) * g.tiles_k, S_eff), dim3(256), sizeof(float) * 4 * 64 * (32 + 4), s, g);
    hipLaunchKernelGGL(conv3_dw_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, S_eff, n, dW);
    AID_LAUNCH_CHECK();
    return AID_OK;
}

}  // namespace aid

using namespace aid;
extern "C" {
// Unit-test entries for the implicit-GEMM convolution (W: [Cout][ldw] with ldw >= 9 Cin a multiple of 4, x: CNHW [Cin][B][Hi][Hi], out / dy: [Cout][B Ho Ho]).
int aid_conv_debug(int mode) {
    if (mode == 10) aid::g_ig_vec4 = 0; else if (mode == 11) aid::g_ig_vec4 = 1; else if (mode == 12) aid::g_dw_tm = 0; else if (mode == 13) aid::g_dw_tm = 1; else if (mode == 14) aid::g_dw_big = 0; else if (mode == 15) aid::g_dw_big = 1; else if (mode == 16) aid::g_dw_big = 2; else aid::g_ig_dbg = mode;
    return AID_OK;
}
int aid_conv3_igemm_tm(const float* Wtm, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, void* stream) {
    return conv3_igemm_ex(Wtm, ldw, x, out, Cin, Cout, B, Hi, Ho, stride, 1, (hipStream_t)stream);
}
int aid_conv3_tapmajor_weights(const float* W, int ldw, float* Wtm, int ldwt, int Cin, int Cout, int flip_transpose, void* stream) {
    AID_REQUIRE(W && Wtm, AID_E_NULL, "aid_conv3_tapmajor_weights: null tensor");
    return flip_transpose ? conv3_flip_weights(W, ldw, Wtm, ldwt, Cin, Cout, (hipStream_t)stream, 1) : conv3_tapmajor_weights(W, ldw, Wtm, ldwt, Cin, Cout, (hipStream_t)stream);
}
int aid_conv3_igemm(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, void* stream) {
    return conv3_igemm(W, ldw, x, out, Cin, Cout, B, Hi, Ho, stride, (hipStream_t)stream);
}
int aid_conv3_igemm_dw(const float* dy, const float* x, float* dW, int ldw, int Cin, int Cout, int B, int Hi, int Ho, int stride, float* slab, int S, void* stream) {
    return conv3_igemm_dw(dy, x, dW, ldw, Cin, Cout, B, Hi, Ho, stride, slab, S, (hipStream_t)stream);
}
int aid_conv3_flip_weights(const float* W, int ldw, float* Wt, int ldwt, int Cin, int Cout, void* stream) {
    AID_REQUIRE(W && Wt, AID_E_NULL, "aid_conv3_flip_weights: null");
    return conv3_flip_weights(W, ldw, Wt, ldwt, Cin, Cout, (hipStream_t)stream, 0);
}
}
