// Row-wise (one wavefront per row) kernels: LayerNorm family, adaLN modulation, time embeddings,
// score head / p_sample, pointwise losses.  All HBM/L2-bound; rows are 0.5-8 KB so a 64-lane wave
// holds a row in registers and reduces with DPP shuffles (no LDS, no atomics).
#pragma once
#include "aid_common.h"

namespace aid {

// out[M,dim] = [sin(t*f), cos(t*f)], f_i = freq_scale * exp(-i ln(1e4)/(half-1)); t scaled by
// `tmul` where the row is on the continuous branch (score_networks.py:126,282-291).
// mode: AID_T_DISCRETE / AID_T_CONTINUOUS / 2 = per-row by value (t<=1 -> continuous).
int sinusoid_embed(const float* t, const float* freq_scale, float tmul, int mode, float* out, int M, int dim, hipStream_t s);

// y = act(LN(x)*gamma+beta) * mask.  gamma/beta may be null (non-affine). stats (optional): [M,2] = mean, rstd.
// y = act(LN(x)) * mask (+ add: same [M, D] layout as y; the decoder's skip connection)
int ln_act(const float* x, int ldx, const float* gamma, const float* beta, int act, const float* mask, float* y, int ldy,
           float* stats, int M, int D, hipStream_t s, const float* add = nullptr);

// y = act(LN(x W^T + b)) * mask + add for K <= 32, D in {256, 512, 1024}, act in {ReLU, SiLU}: one pass (lin / stats optional outputs for a backward pass)
bool lin_smallk_ln_act_ok(int K, int D, int act, const void* a, const void* b, const void* c, const void* d, int ldy);
int lin_smallk_ln_act(const float* x, int ldx, int K, const float* W, const float* bias, const float* gamma, const float* beta, int act,
                      const float* mask, const float* add, float* lin, float* y, int ldy, float* stats, int M, int D, hipStream_t s);

// cond = LN(x)*gamma+beta + temb[row*ld_temb + c] ; sc = silu(cond).   (score_networks.py:145-153 + adaLN's SiLU)
// x_mod > 0: row r reads x row (r % x_mod) (obs embedding shared by every step); temb_div > 0: row r adds temb row (r / temb_div)
// (per-step time embedding), else temb row stride ld_temb (0 = one shared row).
int ln_cond(const float* x, int ldx, const float* gamma, const float* beta, const float* temb, int ld_temb, float* cond,
            float* sc, float* stats, int M, int D, hipStream_t s, int x_mod = 0, int temb_div = 0);
// out[r, :] = a[r % a_rows, :] * mask[r, :]
int mul_rows_mod(const float* a, int a_rows, const float* mask, float* out, size_t R, int D, hipStream_t s);

// y = LN(x)*(1+scale)+shift, scale = mod[row*ldmod + c], shift = mod[row*ldmod + D + c]  (score_networks.py:257-270)
int adaln(const float* x, int ldx, const float* mod, int ldmod, float* y, int ldy, float* stats, int M, int D, hipStream_t s);

// temb = t_sin + time_scale * c  on continuous rows, t_sin elsewhere (score_networks.py:134)
int temb_combine(const float* t_sin, const float* c, const float* time_scale, const float* t, int mode, float* out, int M,
                 int D, hipStream_t s);
// cte input: out[M] = 2t-1 (score_networks.py:130)
int affine1(const float* t, float a, float b, float* out, int M, hipStream_t s);

// score = clamp(raw,-10,10) * mult * (continuous ? sqrt(1/(1e-5+t)) : 1)     (score_networks.py:166-170)
int score_head(const float* raw, const float* mult, const float* t, int mode, float* score, int M, int L, hipStream_t s);

// Optional tail of the reverse-step kernels: block 0's adaLN of the NEXT denoise step with the latent projection folded in (what adaln_lp does as a
// launch of its own), applied to the z row the kernel has just produced - the row is all the next step's first kernel needs, so the two are one launch.
// Wt == nullptr: no tail.  mod = the next step's modulation row block.
struct NextLp { const float* Wt; const float* blp; const float* mod; int ldmod; float* xout; int ldxo; float* y; int ldy; int D; };
#if defined(__HIPCC__)
// x = z W_lp^T + b_lp (Wt = W_lp^T, [L][D]); y = LN(x) (1 + scale) + shift; one wave per row, D = 256 NV.  `zr`: the row's L values (global or LDS).
// The K loop runs in chunks of 8: the 8 z values and 8 NV weight float4s of a chunk are requested together (a K = 17 row was 17 dependent load
// round trips before: 7.5 us of a ~250 us denoise step).
template <int NV>
__device__ __forceinline__ void adaln_lp_row(const float* zr, int L, const float* __restrict__ Wt, const float* __restrict__ blp, const float* __restrict__ modrow,
                                             float* __restrict__ xrow, float* __restrict__ yrow, int lane) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int D = 256 * NV;
    const f4* sc = reinterpret_cast<const f4*>(modrow);
    const f4* sh = reinterpret_cast<const f4*>(modrow + D);
    f4 v[NV], a[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = reinterpret_cast<const f4*>(blp)[lane + 64 * i]; a[i] = sc[lane + 64 * i]; b[i] = sh[lane + 64 * i]; }
    for (int k0 = 0; k0 < L; k0 += 8) {
        float zk[8]; f4 w[8][NV];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = min(k0 + q, L - 1);
            zk[q] = k0 + q < L ? zr[k] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) w[q][i] = reinterpret_cast<const f4*>(Wt + (size_t)k * D)[lane + 64 * i];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)                      // k ascending, as before: the sum order is unchanged
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = k0 + q < L ? fmaf(zk[q], w[q][i][j], v[i][j]) : v[i][j];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<f4*>(xrow)[lane + 64 * i] = v[i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * (1.0f + a[i][j]) + b[i][j];
        reinterpret_cast<f4*>(yrow)[lane + 64 * i] = o;
    }
}
#endif

// fused head + p_sample (diffusion.py:208-255) for one reverse step; coef = coefs + 5*step (device)
//   s = clamp(raw)*mult*tw ; z0 = (z + c0*s)*c1 ; mean = c2*z0 + c3*z ; z' = mean + (step>0 ? c4*eps : 0)
int head2_p_sample(const float* p0, int ldp, const float* W2, int Kp, const float* mult, float t_value, int continuous, const float* coef,
                   int step, const float* z, const float* eps, float* z_out, float* traj_out, float* raw_out, int M, int L, hipStream_t s,
                   const NextLp* next = nullptr);
bool head2_next_ok(int Kp, int L, const NextLp& nx);      // can the reverse-step kernels take this tail? (D == 2 Kp in {512, 1024}, L <= 128, alignment)
int head_p_sample(const float* raw, const float* mult, float t_value, int continuous, const float* coef, int step,
                  const float* z, const float* eps, float* z_out, float* traj_out, int M, int L, hipStream_t s);

// ---- mid-M GEMM of the reverse loop (gemm_mid.hip): 129..1024 rows as exactly one workgroup per CU (8 row tiles), up to MID_MAX_M rows as 128-row
// tiles (3 072 rows = the three merged belief updates of a batch-1024 step: 192 workgroups per 512-wide product); optional split-K into raw slabs
constexpr int MID_MAX_M = 8192;
// K slices for the 512-wide products of a DiT block at M rows: enough (column tile, slice) pairs for ~one workgroup per CU
inline int mid_ksplit(int M) { return M <= 1024 ? 4 : (M <= 2048 ? 2 : 1); }
bool gemm_mid_ok(int M, int N, int K, int ksplit, int act, const float* A, int lda, const float* W, int ldw);           // 129..MID_MAX_M rows
bool gemm_mid_shape_ok(int M, int N, int K, int ksplit, int act, const float* A, int lda, const float* W, int ldw);     // any M >= 1
int gemm_mid(const float* A, int lda, const float* W, int ldw, const float* bias, int act, const float* residual, int ldr, float* C, int ldc,
             float* slab, int M, int N, int K, int ksplit, hipStream_t s);
// general form: w_kmajor = W stored [K][N] (the data gradient dX = dY W of a Linear: the weight as it is), preact = optional second output
int gemm_mid_ex(const float* A, int lda, const float* W, int ldw, int w_kmajor, const float* bias, int act, const float* residual, int ldr, float* C, int ldc,
                float* preact, float* slab, int M, int N, int K, int ksplit, hipStream_t s, const float* dact_pre = nullptr, int dact = 0, int* ctr = nullptr,
                int accumulate = 0);      // ctr: fold the K slices in the kernel (one zeroed int per output tile) and run the full epilogue into C
// <= 128 rows (act() / collector batches, 32-row shards of a strong-scaling run): 32 x 16 tiles, the waves of a workgroup a part of the K slice
// each, operands straight into registers (gemm_mid.hip, gemm_w4_kernel)
bool gemm_w4_ok(int M, int N, int K, int ksplit, int act, const float* A, int lda, const float* W, int ldw);
int gemm_w4(const float* A, int lda, const float* W, int ldw, const float* bias, int act, float* C, int ldc, float* slab, int M, int N, int K, int ksplit,
            hipStream_t s);
// x_new = res + bias + sum of `nslab` slabs [M,D] (fixed order) -> xout (optional);  y = LN(x_new) (1 + scale) + shift (adaLN, mod row = [scale | shift])
// <= 16 rows: gemm_w4 with the adaLN that produces its A operand as a prologue (res + bias_in + slab_in[0..nslab_in) -> LN -> modulation; x_new -> xout once)
bool gemm_w4ln_ok(int M, int N, int K, int ksplit, int act, const float* W, int ldw, int nslab_in);
int gemm_w4ln(const float* res, int ldres, const float* slab_in, int nslab_in, const float* bias_in, const float* mod, int ldmod, float* xout, int ldxo,
              const float* W, int ldw, const float* bias, int act, float* C, int ldc, float* slab, int M, int N, int K, int ksplit, hipStream_t s);
bool adaln_sum_ok(int D, const float* res, int ldres, const float* slab, const float* bias, const float* mod, int ldmod, const float* xout, int ldxo,
                  const float* y, int ldy);
int adaln_sum(const float* res, int ldres, const float* slab, int nslab, const float* bias, const float* mod, int ldmod, float* xout, int ldxo,
              float* y, int ldy, int M, int D, hipStream_t s);

// block 0's adaLN with the latent projection folded in (Wt = W_lp^T, [L][D]); the head's reverse step with its first layer given as split-K slabs
int transpose_small(const float* W, float* Wt, int R, int C, hipStream_t s);
int adaln_lp(const float* z, int L, const float* Wt, const float* blp, const float* mod, int ldmod, float* xout, int ldxo, float* y, int ldy, int M, int D,
             hipStream_t s);
bool head2_slab_ok(int Kp, int nslab, const float* slab, const float* b0, const float* W2);
int head2_slab_p_sample(const float* slab, int nslab, const float* b0, const float* W2, int Kp, const float* mult, float t_value, int continuous,
                        const float* coef, int step, const float* z, const float* eps, float* z_out, float* traj_out, int M, int L, hipStream_t s,
                        const NextLp* next = nullptr);

int fill_const(float* p, float v, size_t n, hipStream_t s);
int mul_elem(const float* a, const float* b, float* out, size_t n, hipStream_t s);
int copy_f32(const float* a, float* out, size_t n, hipStream_t s);

}  // namespace aid
