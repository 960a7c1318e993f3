/* libaid_hip.so — C ABI of the MI355X-native (gfx950) diffusion-policy training hot path.
 *
 * Drop-in boundary for ONE path of neuronphysics/active-inference-diffusion: the
 * reference has no FFI of its own (pure Python/PyTorch), so each entry point below replaces the
 * ATen op chain of one reference method; the reference-side binding is a ctypes stub
 * (INTEGRATION.md).  Citations are `file:line` in the reference, `aid/` = active_inference_diffusion/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major float32 unless stated; all buffers
 *     are BORROWED from the caller (e.g. the PyTorch caching allocator) for the duration of the
 *     call; the library allocates no device memory and keeps no per-call global state;
 *   - `ws`/`ws_bytes` is caller-owned scratch; query its size with the matching *_workspace();
 *   - `stream` is a hipStream_t (passed as void*); nothing synchronises the device;
 *   - return 0 on success, a negative AID_E_* otherwise; aid_last_error() gives the message
 *     (thread-local); nothing throws or aborts across the ABI;
 *   - randomness: every stochastic op takes explicit noise tensors.  Pass NULL to have the
 *     library draw them on device from its counter-based Philox4x32-10 stream (seed, offset).
 */
#ifndef AID_HIP_H
#define AID_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AID_OK 0
#define AID_E_NULL (-1)
#define AID_E_SHAPE (-2)
#define AID_E_WORKSPACE (-3)
#define AID_E_HIP (-4)
#define AID_E_UNSUPPORTED (-5)

#define AID_MAX_BLOCKS 8
#define AID_ABI_VERSION 1

int aid_version(void);
const char* aid_last_error(void);

/* ---- profiling (bench.py's live roofline numbers) ------------------------------------------
 * When enabled every GEMM / optimiser launch is bracketed by HIP events on its own stream.
 * aid_prof_read synchronises those events and returns totals since the last reset:
 * out[0]=launches, out[1]=milliseconds, out[2]=algorithmic FLOP, out[3]=algorithmic bytes. */
int aid_prof_enable(int on);
int aid_prof_reset(void);
int aid_prof_read(int kernel_id, double out[4]);
#define AID_PROF_GEMM 0
#define AID_PROF_ROWWISE 1
#define AID_PROF_OPTIM 2

/* ---- K-level: dense contraction --------------------------------------------------------------
 * C[M,N] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) (+ residual) ; see csrc/gemm_f32.hip.
 * a_kmajor/b_kmajor: 0 = operand stored [rows][K] (k contiguous), 1 = stored [K][rows].
 * act: 0 none, 1 relu, 2 silu, 3 gelu(erf).  Exposed for unit tests and kernel benches. */
int aid_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor, float* C, int ldc,
                 int M, int N, int K, const float* bias, int act, const float* residual, int ldr, float alpha,
                 int accumulate, void* stream);

/* ---- Philox fills (production RNG; parity tests pass explicit tensors instead) ---------------- */
int aid_fill_normal(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);
int aid_fill_uniform(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);
int aid_fill_dropout_mask(float* out, size_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ---- A1: LatentScoreNetwork (aid/models/score_networks.py:12-171) -----------------------------
 * Parameter table: device pointers named after the reference state_dict keys.
 * The SAME struct type with non-const use carries gradient destinations. */
typedef struct AidDiTBlock {
    float *ada1_w, *ada1_b;         /* norm1.adaLN_modulation.1  [2D,D],[2D] */
    float *ada2_w, *ada2_b;         /* norm2.adaLN_modulation.1 */
    float *in_proj_w, *in_proj_b;   /* attention.in_proj_{weight,bias} [3D,D],[3D] (rows 2D.. live) */
    float *out_proj_w, *out_proj_b; /* attention.out_proj [D,D],[D] */
    float *mlp0_w, *mlp0_b;         /* mlp.0 [4D,D],[4D] */
    float *mlp2_w, *mlp2_b;         /* mlp.2 [D,4D],[D] */
} AidDiTBlock;

typedef struct AidScoreNet {
    int32_t L, O, D, TE, n_blocks, _pad;
    float *time_scale, *output_multiplier, *te_freq_scale;            /* [], [1], [1] */
    float *te1_w, *te1_b, *te3_w, *te3_b;                             /* time_embed.{1,3} */
    float *oe0_w, *oe0_b, *oe1_g, *oe1_b;                             /* obs_encoder.{0,1} */
    float *oe4_w, *oe4_b, *oe5_g, *oe5_b;                             /* obs_encoder.{4,5} */
    float *oe7_w, *oe7_b, *oe8_g, *oe8_b;                             /* obs_encoder.{7,8} */
    float *cte0_w, *cte0_b, *cte2_w, *cte2_b, *cte4_w, *cte4_b;       /* continuous_time_embed */
    float *lp_w, *lp_b;                                               /* latent_proj */
    AidDiTBlock blk[AID_MAX_BLOCKS];
    float *adaf_w, *adaf_b;                                           /* norm_final.adaLN_modulation.1 */
    float *op0_w, *op0_b, *op2_w;                                     /* output_proj.{0,2} */
} AidScoreNet;

/* time branch of score_networks.py:121 decided by the CALLER (no device->host sync):
 *   AID_T_CONTINUOUS: every t in [0,1]  (ELBO; reverse-loop steps 0 and 1, reference quirk D8)
 *   AID_T_DISCRETE  : raw-t embedding, no output time weight. */
#define AID_T_DISCRETE 0
#define AID_T_CONTINUOUS 1

/* s_theta(z_t, t, o) -> score[B,L].  t: [B] floats.  drop_mask: [B,D] scaled keep mask of the obs
 * encoder's Dropout(0.1) or NULL for eval mode. */
size_t aid_score_forward_workspace(const AidScoreNet* net, int B);
int aid_score_forward(const AidScoreNet* net, int B, const float* z_t, const float* t, int t_mode,
                      const float* obs, const float* drop_mask, float* score, void* ws, size_t ws_bytes,
                      void* stream);

/* ---- A2/A3/A4: reverse-diffusion loop (aid/core/diffusion.py:176-255) ------------------------
 * coefs: device [N,5] table (sqrt(1-acp), 1/sqrt(alpha), coef1, coef2, sqrt(post_var)).
 * z_init: [B,L] N(0,1) draw; step_noise: [N,B,L] (row t used at step t>0); drop_masks: [N,B,D]
 * indexed by step t, or NULL (eval mode).  NULL z_init/step_noise/drop_masks(with train=1) =>
 * Philox draws.  traj (optional): [N+1,B,L] receives every iterate.  Output z: [B,L]. */
size_t aid_denoise_loop_workspace(const AidScoreNet* net, int B, int n_steps);
int aid_denoise_loop(const AidScoreNet* net, int B, int n_steps, const float* coefs, const float* obs,
                     const float* z_init, const float* step_noise, const float* drop_masks, int train,
                     uint64_t seed, uint64_t offset, float* z_out, float* traj, void* ws, size_t ws_bytes,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AID_HIP_H */
