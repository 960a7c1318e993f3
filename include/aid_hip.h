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
/* Bumped whenever an existing entry point changes its signature or a table struct its layout (additions do not bump it).
 *   2: aid_value_step / aid_value_step_ex take `lam` as double (round 3; was float). */
#define AID_ABI_VERSION 2

int aid_version(void);
const char* aid_last_error(void);

/* ---- profiling (bench.py's live roofline numbers) ------------------------------------------
 * When enabled every GEMM / optimiser launch is bracketed by HIP events on its own stream.
 * aid_prof_read synchronises those events and returns totals since the last reset:
 * out[0]=launches, out[1]=milliseconds, out[2]=algorithmic FLOP, out[3]=algorithmic bytes. */
int aid_prof_enable(int on);
int aid_prof_reset(void);
int aid_prof_read(int kernel_id, double out[4]);
/* One CSV line per recorded launch (kernel_id,M,N,K,layout,milliseconds,variant): the in-situ per-shape timing the GEMM dispatcher is tuned
 * from.  layout = 2*a_kmajor + b_kmajor (+4 in slab split-K mode); variant = the aid_gemm_force_kernel code of the kernel that ran
 * (1 panel, 2 ring, 3 small-K, 4/5 tile (vector / scalar loads), 6/7 split-K (vector / scalar loads), 8 thin TN, 10 skinny (M <= 32), 9/11 big, 14 panel128, 15 panel64): with the layout it
 * names one template instantiation, i.e. one row of a rocprofv3 --stats summary. */
int aid_prof_dump(const char* path);
#define AID_PROF_GEMM 0        /* GEMM launches that ran a small-problem kernel (panel / split-K / ring / skinny / thin / small-K) */
#define AID_PROF_GEMM_TILE 3   /* GEMM launches that ran a 64x64-tile kernel (gemm_big_kernel, the dominant kernel of a train_step, or gemm_tile_kernel) */
#define AID_PROF_ROWWISE 1
#define AID_PROF_OPTIM 2

/* ---- K-level: dense contraction --------------------------------------------------------------
 * C[M,N] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) (+ residual) ; see csrc/gemm_f32.hip.
 * a_kmajor/b_kmajor: 0 = operand stored [rows][K] (k contiguous), 1 = stored [K][rows].
 * act: 0 none, 1 relu, 2 silu, 3 gelu(erf).  Exposed for unit tests and kernel benches. */
int aid_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor, float* C, int ldc,
                 int M, int N, int K, const float* bias, int act, const float* residual, int ldr, float alpha,
                 int accumulate, void* stream);

/* The same, with scratch (aid_gemm_scratch_bytes() bytes, borrowed for the call): problems of 129..1024 rows x <= 1024 columns with K >= 1024
 * (the DiT blocks' 512-wide products in aid_elbo_step's passes) are then cut into K slices over ~256 workgroups; the slice that arrives last
 * at an output tile folds the others in slice order and applies the epilogue (csrc/gemm_mid.hip), so the result is complete when the kernel
 * ends and identical run to run.  Unit-test entry for that route; aid_elbo_step takes the scratch from its workspace. */
int aid_gemm_f32_scratch(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor, float* C, int ldc, int M, int N, int K,
                         const float* bias, int act, const float* residual, int ldr, float alpha, int accumulate, void* scratch, size_t scratch_bytes,
                         void* stream);
size_t aid_gemm_scratch_bytes(void);

/* y[M,N] = act(LayerNorm(x W^T + b) * gamma + beta) * mask + add in ONE kernel (full-row tiles, LayerNorm in the GEMM epilogue; csrc/gemm_rowln.hip):
 * N in {512, 1024}, M >= 2048, K >= 64 and a multiple of 4, 16-B aligned operands, act in {0 none, 1 relu, 2 silu}; mask [M,N] (row stride N) / add / gamma / beta / b
 * may be NULL.  What Linear -> LayerNorm -> SiLU stages of the observation decoder run as on the estimator's sample rows
 * (core/active_inference.py:237-242, 940-993).  Unit-test / bench entry. */
int aid_linear_ln_act(const float* x, int ldx, const float* W, const float* b, const float* gamma, const float* beta, int act,
                      const float* mask, const float* add, int ldadd, float* y, int ldy, int M, int N, int K, void* stream);

/* Mid-M GEMM of the reverse loop (csrc/gemm_mid.hip; 128 < M <= 1024, one workgroup per CU): ksplit == 1: C = act(A W^T + bias) + residual
 * (act: 0 none, 1 relu, 2 silu, 3 gelu);  ksplit > 1 (K % (64 ksplit) == 0): slab[y][M][N] = raw partial product of K slice y (bias / act /
 * residual belong to the consumer).  aid_adaln_sum is that consumer for the DiT blocks (models/score_networks.py:214-235, 257-270):
 * x_new = res + bias + sum_y slab[y] in a fixed order -> xout (optional), y = LayerNorm(x_new) (1 + mod[:, :D]) + mod[:, D:].
 * Unit-test / bench entries; the reverse loop calls them through aid_denoise_loop. */
int aid_gemm_mid(const float* A, int lda, const float* W, int ldw, const float* bias, int act, const float* residual, int ldr, float* C, int ldc,
                 float* slab, int M, int N, int K, int ksplit, void* stream);
/* <= 128 rows (csrc/gemm_mid.hip, gemm_w4_kernel): C [M,N] = act(A W^T + bias) (ksplit == 1) or ksplit raw K-slice slabs [ksplit][M][N]; a workgroup owns a
 * 32 x 16 tile, its four or eight waves a part of the K slice each, operands go straight from global memory into the MFMA registers.  K / ksplit in
 * {256, 512, 1024}; act in {none, silu, gelu} (ksplit == 1 only).  The products of the act() / collector reverse loop.  Unit-test entry. */
int aid_gemm_w4(const float* A, int lda, const float* W, int ldw, const float* bias, int act, float* C, int ldc, float* slab, int M, int N, int K, int ksplit,
                void* stream);
/* timing experiments (process-wide, outside the product path; tools/perf): force the row-tile height of the mid-M kernel to >= 32 bmt_min rows (1 = automatic);
 * aid_denoise_debug(n > 0): cap the number of denoise steps that share one hoisted conditioning pass (0 = automatic); -1 / -2: the <= 128-row loop on the
 * gemm_w4 pipeline (default) / on the weight-streaming skinny kernels; -5 / -6: the next step's adaLN as the reverse-step kernel's tail on (default) / off;
 * aid_conv_debug: 10 / 11 four-pixel vector gathers of the tap-major convolutions off / on (default), 12 / 13 the tap-major weight-gradient kernel off / on (default), 22 no 128 x 128 tiles, 1..4 ablation variants of the non-tap-major 128 x 64 tile
 * (results wrong by construction, tools/perf/conv_ablation.py), 0 = normal; aid_rowln_debug(40 / 41): the estimator's 1024-wide stage as co-resident 512-column
 * blocks + a LayerNorm pass on / off (default), other codes see csrc/gemm_rowln.hip */
int aid_mid_debug(int bmt_min);
int aid_denoise_debug(int chunk_cap);
int aid_conv_debug(int mode);
int aid_rowln_debug(int mode);
/* Arithmetic of the estimator's full-row GEMMs (>= 2 048 rows x 512 / 1024 columns): 0 (default) v_mfma_f32_32x32x2_f32; 1: each float32 operand split into
 * three bfloat16 values (exactly: 8 + 8 + 8 bits) and the six leading partial products accumulated in float32 on v_mfma_f32_32x32x16_bf16 - closer to the
 * float64 product than the float32 MFMA chain (csrc/gemm_rowln_x6.hip, tools/perf/native/bf16x6_probe.hip), 2.67 x the MFMA rate per clock.  Process-wide;
 * environment AID_GEMM_X6=1 sets the initial value. */
int aid_gemm_emulation(int mode);
int aid_gemm_emulation_mode(void);      /* the current value (1 / 0) */
int aid_adaln_sum(const float* res, int ldres, const float* slab, int nslab, const float* bias, const float* mod, int ldmod, float* xout, int ldxo,
                  float* y, int ldy, int M, int D, void* stream);

/* C[M,N] += A[M,K] W[N,K]^T + bias with K split over `ksplit` workgroups that atomically add into C (C holds the residual).
 * The order of the float adds is not reproducible run to run (last-bit differences).  Bench/test entry. */
int aid_gemm_ksplit_add(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, const float* bias,
                        int ksplit, void* stream);
/* Implicit-GEMM 3x3 convolution, pad 1, stride 1 or 2, on channel-major ("CNHW") activations - what aid_encoder_forward / _backward and
 * aid_conv_decoder_forward run for every 3x3 conv (aid/encoder/visual_encoders.py:135-189, 228-459); unit-test entries.
 *   x [Cin][B][Hi][Hi], W [Cout][ldw] (ldw >= 9 Cin, multiple of 4, k = ci 9 + kh 3 + kw), out / dy [Cout][B Ho Ho]
 *   aid_conv3_igemm      out = conv(x, W)
 *   aid_conv3_flip_weights  Wt[ci][co 9 + kh 3 + kw] = W[co][ci 9 + (2-kh) 3 + (2-kw)]: dx = aid_conv3_igemm(Wt, dy) with Cin <-> Cout (stride 1)
 *   aid_conv3_igemm_dw   dW[co][k] = sum_n dy[co][n] X(k, n); slab: scratch [S][Cout][ldw]; B Ho Ho must be a multiple of 4 */
int aid_conv3_igemm(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, void* stream);
/* The same convolution with the contraction in TAP-MAJOR order, k' = (kh 3 + kw) Cin + ci (what the encoder / ConvDecoder run whenever Cin % 16 == 0, % 32 for
 * Cout <= 32: a K stage is then whole channels of one tap and the operand loader needs no per-element tap arithmetic).  Wtm [Cout][ldw] holds the weights in that
 * order: aid_conv3_tapmajor_weights(W, ...) with flip_transpose = 0 (forward) or 1 (Wt[ci][tap Cout + co] = W[co][ci 9 + (8 - tap)]: the data gradient,
 * Cin <-> Cout as for aid_conv3_flip_weights). */
int aid_conv3_igemm_tm(const float* Wtm, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, void* stream);
int aid_conv3_tapmajor_weights(const float* W, int ldw, float* Wtm, int ldwt, int Cin, int Cout, int flip_transpose, void* stream);
int aid_conv3_igemm_dw(const float* dy, const float* x, float* dW, int ldw, int Cin, int Cout, int B, int Hi, int Ho, int stride, float* slab, int S, void* stream);
int aid_conv3_flip_weights(const float* W, int ldw, float* Wt, int ldwt, int Cin, int Cout, void* stream);

/* n weight-gradient problems dW_i[N_i, K_i] += dY_i[M_i, N_i]^T X_i[M_i, K_i] (row-major, dense), and db_i[N_i] += column sums of dY_i when
 * db / db[i] is non-null, in ONE launch (csrc/gemm_group.hip: what the backward passes of aid_elbo_step / aid_efe_step do with all their
 * parameter gradients).  Two consecutive problems with the same dW are contracted into the same accumulators.  scratch (optional, scratch_floats
 * floats): with it, one-segment problems over >= 4 096 rows are cut into up to 8 row chunks whose partial tiles a second launch folds in
 * chunk order (the BPTT's contractions over H*K*B rows).  Unit-test / bench entry. */
int aid_gemm_group_tn(int n, float* const* dW, float* const* db, const float* const* dy, const float* const* x, const int* M, const int* N, const int* K,
                      float* scratch, size_t scratch_floats, void* stream);
/* TESTS / BENCHES ONLY - not part of the product path, process-wide and not thread-safe: kernel-selection override.  0 auto, 1 panel32,
 * 2 ring64, 4 tile64, 6 splitk, 8 thin TN (weight gradients of narrow layers), 9 / 11 gemm_big_kernel with prefetch distance 2 / 1,
 * 10 skinny (M <= 32 rows: act() / collector batches), 14 panel128, 15 panel64; bit 8 selects the XCD-grouped tile map of gemm_big_kernel. */
int aid_gemm_force_kernel(int k);
/* Dispatch hints are per STREAM (the library keeps no per-call global state): flags bit 0 = launches on `stream` prefer the 35 KB-LDS
 * small-problem GEMM variant, so that they co-reside with large GEMMs another stream is running (speed only, results identical); bit 1 = the
 * 512-wide full-row kernels launched on `stream` take 16-deep stages (74 KB of LDS per workgroup instead of 147), so that another stream's
 * 80 KB workgroups (the reverse loop's) can share their CUs.  Set it once per stream; flags = 0 clears.  Launches on every other stream - e.g. a collector thread's inference stream
 * (utils/async_collector.py:388,451) - are unaffected.  Thread-safe. */
int aid_stream_hint(void* stream, int flags);

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
 * encoder's Dropout(0.1) or NULL for eval mode.  obs == NULL: unconditional scoring, the zero observation
 * embedding of aid/models/score_networks.py:144-149 (the obs-encoder parameters are not read). */
size_t aid_score_forward_workspace(const AidScoreNet* net, int B);
int aid_score_forward(const AidScoreNet* net, int B, const float* z_t, const float* t, int t_mode,
                      const float* obs, const float* drop_mask, float* score, void* ws, size_t ws_bytes,
                      void* stream);

/* ---- Sub-module forwards of LatentScoreNetwork as stand-alone operations (inference; the training / sampling paths use the fused entry points above).
 *   aid_sinusoid_embed        SinusoidalPositionEmbeddings.forward  aid/models/score_networks.py:282-291   out [M, dim] = [sin(t f_i) | cos(t f_i)]
 *   aid_adaptive_layer_norm   AdaptiveLayerNorm.forward             :257-270   y = LN(x) (1 + scale) + shift, [scale | shift] = SiLU(cond) mod_w^T + mod_b
 *   aid_dit_block_forward     DiTBlock.forward                      :214-235   (single-token attention = out_proj(V_proj(.)): rows 2D.. of in_proj)
 * Workspace of the last two: aid_module_workspace(M, D). */
int aid_sinusoid_embed(const float* t, const float* freq_scale, int M, int dim, float* out, void* stream);
size_t aid_module_workspace(int M, int D);
int aid_adaptive_layer_norm(const float* x, const float* cond, const float* mod_w, const float* mod_b, int M, int D, float* y, void* ws, size_t ws_bytes,
                            void* stream);
int aid_dit_block_forward(const AidDiTBlock* blk, int M, int D, const float* x, const float* cond, float* y, void* ws, size_t ws_bytes, void* stream);
/* SpatialAttention.forward (aid/encoder/visual_encoders.py:210-225) on an NCHW tensor: out = x + x sigmoid(conv7x7([mean_c x, max_c x]) / temperature).
 * conv_w [2*7*7], conv_b [1], temperature [1] device pointers.  Workspace: 3 B H W floats (+ alignment). */
int aid_spatial_attention_forward(const float* x, const float* conv_w, const float* conv_b, const float* temperature, int B, int C, int H, int W, float* out,
                                  void* ws, size_t ws_bytes, void* stream);

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


/* A6: AsyncGPUInference._batch_diffusion_sampling (aid/utils/async_collector.py:530-595): `num_steps` <= max_index+1 reverse
 * steps; the score net is fed t = step/max_index (always the continuous branch), p_sample the integer step; after every step a
 * whole-batch NaN/Inf test re-initialises z = 0.1*N(0,1) (reinit_noise [num_steps,B,L] or NULL = Philox; reinit_count, optional
 * device int, counts the re-initialisations).  Other arguments as aid_denoise_loop.  Workspace: aid_denoise_loop_workspace. */
int aid_collector_sampling(const AidScoreNet* net, int B, int num_steps, int max_index, const float* coefs, const float* obs,
                           const float* z_init, const float* step_noise, const float* drop_masks, const float* reinit_noise,
                           int train, uint64_t seed, uint64_t offset, float* z_out, int* reinit_count, void* ws, size_t ws_bytes,
                           void* stream);

/* ---- small networks (A4/A9/A10): parameter tables named after the reference modules -----------
 * aid/models/policy_networks.py:42-71, value_networks.py:24-45, dynamics_models.py:28-45,
 * aid/core/active_inference.py:111-131 (decoder), :164-171 (reward predictor), :886-919 (MINE). */
typedef struct AidLinear { float *w, *b; int32_t in, out; } AidLinear;  /* torch Linear: w [out,in], b [out] (may be NULL) */
typedef struct AidLN { float *g, *b; } AidLN;                           /* LayerNorm weight / bias */

typedef struct AidPolicyNet {
    int32_t L, A, D, _pad;
    AidLinear enc0; AidLN enc_ln; AidLinear enc3;        /* latent_encoder.{0,1,3} */
    AidLinear trunk[3]; AidLN trunk_ln[3];               /* trunk.{0,3,6} / {1,4,7} */
    AidLinear mean0, mean2, lstd0, lstd2;                /* mean_head.{0,2}, log_std_head.{0,2} */
} AidPolicyNet;
typedef struct AidValueNet {
    int32_t L, D, TE, _pad;
    float* freq_scale;                                   /* time_embed.0.freq_scale */
    AidLinear te1;                                       /* time_embed.1 */
    AidLinear net[3]; AidLN ln[3]; AidLinear out;        /* network.{0,3,6}/{1,4,7}/9 */
} AidValueNet;
typedef struct AidDynamicsNet {
    int32_t L, A, D, _pad;
    AidLinear net[3]; AidLN ln[3]; AidLinear out;
} AidDynamicsNet;
typedef struct AidRewardNet { int32_t L, D; AidLinear l0; AidLN ln; AidLinear l3, l5; } AidRewardNet;
typedef struct AidDecoderNet { int32_t L, O, D, _pad; AidLinear s[3]; AidLN ln[3]; AidLinear out; } AidDecoderNet;
typedef struct AidEpistemicNet {
    int32_t L, O;
    float* perturbation_scale;
    AidLinear fe0, fe2, fe4;                             /* feature_extractor.{0,2,4} */
    AidLinear jp0; AidLN jp_ln; AidLinear jp4;           /* jacobian_projector.{0,1,4} */
    AidLinear lp0, lp2;                                  /* latent_processor.{0,2} */
    AidLinear mn0, mn3, mn6;                             /* mine_network.{0,3,6} */
} AidEpistemicNet;
typedef struct AidDiffusionParams { float *prior_mean, *prior_log_std, *log_snr_min, *log_snr_max; } AidDiffusionParams;

/* ---- ConvDecoder (pixel-mode observation decoder), forward / eval mode -------------------------------------------------------
 * active_inference_diffusion/encoder/visual_encoders.py:228-459.  Spectral-normed convs carry weight_orig + u + v (legacy layout);
 * plain convs have u = v = NULL. */
typedef struct AidSNConv { const float *w, *u, *v, *bias; } AidSNConv;
typedef struct AidDecoderBlock { AidSNConv main0, main1, res; } AidDecoderBlock;     /* res.w == NULL: identity residual */
typedef struct AidConvDecoder {
    int32_t L, hidden;
    AidLinear l0; AidLN ln1; AidLinear l4; AidLN ln5;                    /* latent_proj.{0,1,4,5} */
    AidDecoderBlock blk[4];
    const float *out0_w, *out0_b, *out3_w, *out3_b, *out6_w, *out6_b;    /* output_proj.{0,3,6} */
} AidConvDecoder;
/* Pixel branch of FunctionSpaceEpistemicEstimator (active_inference.py:865-925; utils/util.py:7-75), minus its alias of the decoder. */
typedef struct AidPixelEpistemicNet {
    int32_t L, _pad;
    float* perturbation_scale;
    const float *pp0_w, *pp0_b, *pp2_w, *pp2_b, *pp4_w, *pp4_b;          /* pixel_processor.{0,2,4}: 5x5 stride-2 convs */
    const float *pos, *queries, *in_proj_w, *in_proj_b;                  /* spatial_aggregator: pos_encoding [441,128], epistemic_queries [16,128], MHA in_proj */
    AidLinear att_out, agg0; AidLN agg_ln; AidLinear agg3;               /* attention.out_proj, output_proj.{0,1,3} */
    AidLinear jp0; AidLN jp_ln; AidLinear jp4;                           /* jacobian_projector.{0,1,4} (in = 4 * 256) */
    AidLinear lp0, lp2, mn0, mn3, mn6;                                   /* latent_processor, mine_network */
} AidPixelEpistemicNet;
/* z [R, L] -> out [R, 3, 168, 168] (NCHW, in [0,1]); R <= 64 per call (callers chunk larger batches). */
size_t aid_conv_decoder_workspace(const AidConvDecoder* dec, int R);
int aid_conv_decoder_forward(const AidConvDecoder* dec, int R, const float* z, float* out, void* ws, size_t ws_bytes, void* stream);

/* decode_observation, state branch (active_inference.py:237-242).  masks: 3 scaled keep masks
 * [B,2D],[B,2D],[B,D] (Dropout 0.2) or NULL pointers for eval mode. */
size_t aid_decode_workspace(const AidDecoderNet* dec, int B);
int aid_decode_observation(const AidDecoderNet* dec, int B, const float* z, const float* mask0, const float* mask1,
                           const float* mask2, float* out, void* ws, size_t ws_bytes, void* stream);

/* A4 update_belief_via_diffusion (active_inference.py:256-312): reverse loop + batch stats + decode + MSE.
 * stats_out: [2L+1] = latent mean[L], unbiased std[L], reconstruction error.  dec_masks: 3 pointers or NULL. */
size_t aid_belief_update_workspace(const AidScoreNet* net, const AidDecoderNet* dec, int B, int n_steps);
int aid_belief_update(const AidScoreNet* net, const AidDecoderNet* dec, int B, int n_steps, const float* coefs,
                      const float* obs, const float* z_init, const float* step_noise, const float* drop_masks,
                      const float* const* dec_masks, int train, uint64_t seed, uint64_t offset, float* z_out,
                      float* stats_out, void* ws, size_t ws_bytes, void* stream);

/* ---- A5/A7: compute_diffusion_elbo forward + backward (active_inference.py:533-636, 709-771) ---
 * Gradients of loss = -elbo are ACCUMULATED into `grads` / `dgrads` (caller zeroes them): score net incl. the
 * gradient-penalty double backward, log_snr_min/max, latent prior.  Decoder / reward-predictor gradients are
 * not produced (the reference zeroes them before any optimiser step: state_agent.py:225).
 * noise: any NULL member is drawn on device (Philox, seed/offset).  t NULL => uniform (first call) or importance
 * sampled from time_importance[100] (later calls).  time_importance is updated in place (sequential EMA, K10).
 * metrics (device, 16 floats): 0 recon, 1 kl, 2 score_matching, 3 elbo, 4 reward_nll, 5 grad_penalty,
 * 6 mean_time, 7 loss_weight_mean, 8 loss(=-elbo). */
typedef struct AidElboNoise {
    const float *t, *eps, *prior_eps, *score_mask, *gp_mask, *dec_mask0, *dec_mask1, *dec_mask2;
    float* d_obs_out;   /* optional OUTPUT [B,O]: accumulates d(loss)/d(observations) from the score net's obs encoder (main + GP pass);
                           pixel mode, where the observations are the DrQ-v2 encoder's features.  With dec == NULL the reconstruction
                           term is taken from metrics[0] as pre-set by the caller (aid_feature_decoder_step). */
    const float* mean_t;/* optional device scalar: mean of t over the GLOBAL batch.  The KL annealing weight exp(-5 mean t)
                           (active_inference.py:611) is a batch statistic; a data-parallel caller draws t first (aid_sample_time),
                           all-gathers it and passes the global mean so that N ranks x B rows equal one N*B-row step.  NULL => mean
                           over this call's B rows. */
} AidElboNoise;
typedef struct AidElboCfg {
    float kl_weight, diffusion_weight, reward_weight, gp_weight;
    int32_t train, first_call;
    uint64_t seed, offset;
    int32_t defer_time_importance, _pad;   /* 1: leave time_importance untouched (data-parallel: caller applies the
                                              sequential EMA over the all-gathered (t, loss) pairs in global order) */
} AidElboCfg;
/* the sequential time-importance EMA of active_inference.py:750-771 over n (t, per-sample loss) pairs, in order */
int aid_time_importance_update(float* time_importance, const float* t, const float* per_sample, int n, int first_call, void* stream);
/* _importance_sample_time (active_inference.py:731-748) alone: t_out [B] ~ uniform (first_call) or importance-sampled from
 * time_importance[100], device Philox (seed, offset).  ws: >= 2*B floats. */
int aid_sample_time(const float* time_importance, int first_call, int B, uint64_t seed, uint64_t offset, float* t_out, void* ws,
                    size_t ws_bytes, void* stream);
size_t aid_elbo_step_workspace(const AidScoreNet* net, const AidDecoderNet* dec, int B);

/* DiffusionActiveInference._compute_gradient_penalty (aid/core/active_inference.py:709-729) as a stand-alone value:
 * g = d(sum score(z, t, obs))/dz, penalty = mean_b (||g_b||_2 - 1)^2 -> penalty_out (device scalar); grad_out (optional) [B,L] receives g.
 * The same forward + input-gradient pass aid_elbo_step runs for its metric 5 (which also back-propagates the penalty: that part
 * stays inside aid_elbo_step).  drop_mask: [B,D] scaled keep mask of the obs encoder's Dropout(0.1) or NULL (eval).
 * Workspace: aid_elbo_step_workspace(net, NULL, B). */
int aid_gradient_penalty(const AidScoreNet* net, int B, const float* z, const float* t, int t_mode, const float* obs, const float* drop_mask,
                         float* penalty_out, float* grad_out, void* ws, size_t ws_bytes, void* stream);

/* LatentScoreNetwork as a differentiable op (what `loss.backward()` does for a loss built on score_network(z, t, obs): aid/models/score_networks.py:101-171 under
 * autograd): forward with the activations kept, then the reverse pass for the cotangent d_score [B,L] = dLoss/dscore.  Parameter gradients are ACCUMULATED into
 * `grads` (same table type; zero it first for a fresh gradient); score_out (optional) [B,L] = the forward value; dz_out (optional) [B,L] = dLoss/dz;
 * d_obs_out (optional) [B,O] ACCUMULATES dLoss/dobs.  drop_mask: [B,D] scaled keep mask of the obs encoder's Dropout(0.1) or NULL (eval).
 * Workspace: aid_elbo_step_workspace(net, NULL, B). */
int aid_score_backward(const AidScoreNet* net, const AidScoreNet* grads, int B, const float* z, const float* t, int t_mode, const float* obs, const float* drop_mask,
                       const float* d_score, float* score_out, float* dz_out, float* d_obs_out, void* ws, size_t ws_bytes, void* stream);

/* F.mse_loss(a, b) over [M,N] row-major operands with leading dimensions lda / ldb -> out (device scalar): the reconstruction
 * diagnostic of update_belief_via_diffusion's pixel branch (aid/core/active_inference.py:292-299). */
int aid_mse(const float* a, int lda, const float* b, int ldb, int M, int N, float* out, void* stream);
int aid_elbo_step(const AidScoreNet* net, const AidScoreNet* grads, const AidDiffusionParams* dp,
                  const AidDiffusionParams* dgrads, const AidDecoderNet* dec, const AidRewardNet* rew, int B,
                  const float* obs, const float* rewards, const float* latents, const AidElboNoise* noise,
                  const AidElboCfg* cfg, float* time_importance, float* t_out, float* per_sample_out,
                  float* metrics, void* ws, size_t ws_bytes, void* stream);

/* ---- K15: clip_grad_norm_ -> AdamW -> EMA over one flat parameter arena (state_agent.py:152-158) ---
 * p,g,m,v,(ema) are [n] float32 arenas.  All on device: the global L2 norm (deterministic two-stage
 * reduction) lands in scratch[0], the clip coefficient min(1, clip/(norm+1e-6)) in scratch[1].
 * step is 1-based.  ema may be NULL.  scratch: >= 4096 floats. */
int aid_adamw_step(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay,
                   float beta1, float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch,
                   void* stream);


/* ---- A8/A10: compute_expected_free_energy_diffusion + MINE epistemic estimator, forward + policy backward
 * (active_inference.py:314-418, 940-1063).  The K trajectories are batched: row r = k*B + b.
 * Noise tensors (NULL member => Philox draw), H = horizon, M = K*B, R = K*S*B:
 *   policy_eps [H,M,A]  rsample draws;  reparam_eps [H,M,L]  reparameterize draws;
 *   mine_z_eps [H,R,L] (row (k*S+s)*B+b);  mine_delta [4,H,R,L] raw N(0,1) perturbation directions (direction j outermost);
 *   proj_mask [H,R,512] jacobian_projector dropout;  mine_mask [4,H,R,512] = joint(0,1), marginal(2,3) masks (mask kind outermost: the
 *   estimator runs ONCE over all H horizon steps, H*R rows, so each per-layer operand is one contiguous [H*R, .] matrix);
 *   perms [H,K,S,B] int32 permutations (randperm per sample block).
 * Outputs: efe[B]; info (device, 16 floats): 0 policy_loss(=mean efe) 1 epistemic_mean 2 pragmatic_mean
 * 3 consistency_mean 4 mi_estimate 5 joint_term 6 marginal_term (last (k,t) call) ;
 * running_mean: device scalar, EMA updated in the reference's (k outer, t inner) order.
 * pgrads == NULL => forward only (act()).  Only POLICY gradients are produced: the EFE is minimised by the
 * policy optimiser alone (state_agent.py:162-177); data gradients flow through dynamics/reward/value. */
typedef struct AidEfeNoise {
    const float *policy_eps, *reparam_eps, *mine_z_eps, *mine_delta, *proj_mask, *mine_mask;
    const int32_t* perms;
} AidEfeNoise;
typedef struct AidEfeCfg {
    int32_t H, K, S, train, skip_epistemic, _pad;
    float epistemic_weight, pragmatic_weight, consistency_weight, discount, temperature, mine_alpha;
    uint64_t seed, offset;
} AidEfeCfg;
/* DiffusionActiveInference.train_epistemic_estimator (active_inference.py:420-445): one forward + backward of the MINE estimator on
 * S samples of N(predict_next_latent(latents, actions), 0.1): loss = -(mean t_joint - EMALoss(t_marginal)).  Gradients of every
 * estimator-owned parameter are ACCUMULATED into `grads` (same table type; the decoder gets none - it runs under no_grad in the
 * reference); running_mean (device scalar) gets its EMA update; out4 = {mi, joint, marginal_term, loss}.  noise: mine_z_eps [S*B,L],
 * mine_delta [4,S*B,L], proj_mask [S*B,512], mine_mask [4,S*B,512], perms [S,B] (NULL members => device Philox from seed/offset). */
size_t aid_mine_train_workspace(const AidDecoderNet* dec, const AidDynamicsNet* dyn, int B, int S);
int aid_mine_train_step(const AidDynamicsNet* dyn, const AidDecoderNet* dec, const AidEpistemicNet* epi, const AidEpistemicNet* grads, int B, int S,
                        const float* latents, const float* actions, const AidEfeNoise* noise, int train, uint64_t seed, uint64_t offset,
                        float alpha, float* running_mean, float* out4, void* ws, size_t ws_bytes, void* stream);
/* DiffusionActiveInference.compute_epistemic_value (active_inference.py:398-418) = one gradient-free forward of the state-mode
 * FunctionSpaceEpistemicEstimator (:995-1063): S samples of N(next_mean, exp(next_logvar)) per row, finite-difference decoder features,
 * MINE statistics on joint / row-permuted marginal.  next_mean, next_logvar [B,L].  noise members as aid_mine_train_step (NULL => device
 * Philox).  running_mean (device scalar) gets the EMA update the reference applies even under no_grad.
 * out4 = {mi, joint term, marginal term, -clamp(mi, 0)}. */
size_t aid_epistemic_value_workspace(const AidDecoderNet* dec, int B, int S);
int aid_epistemic_value(const AidDecoderNet* dec, const AidEpistemicNet* epi, int B, int S, const float* next_mean, const float* next_logvar,
                        const AidEfeNoise* noise, int train, uint64_t seed, uint64_t offset, float alpha, float* running_mean, float* out4,
                        void* ws, size_t ws_bytes, void* stream);
/* Same as aid_efe_step with one more output: next_means_out [H][K*B][L] (or NULL), the dynamics means the estimator samples around. */
int aid_efe_step_ex(const AidPolicyNet* pol, const AidPolicyNet* pgrads, const AidDynamicsNet* dyn, const AidRewardNet* rew,
                    const AidValueNet* val, const AidDecoderNet* dec, const AidEpistemicNet* epi, int B, const float* latents,
                    const AidEfeNoise* noise, const AidEfeCfg* cfg, float* running_mean, float* efe_out, float* info, float* next_means_out,
                    void* ws, size_t ws_bytes, void* stream);
/* DecoderBlock.forward, eval mode (aid/encoder/visual_encoders.py:370-459) as a call of its own: x [R, cin, h, h] -> out [R, cout, ho, ho] (ho = 2h when
 * `upsample`: sub-pixel conv + PixelShuffle(2) on both paths).  blk->res.w == NULL: identity residual (needs cin == cout, no upsampling).  Spectral-norm
 * weights are W / (u . W v) of the stored vectors (eval mode: no power iteration); Dropout2d is the identity in eval mode. */
size_t aid_decoder_block_workspace(int cin, int cout, int upsample, int R, int h);
int aid_decoder_block_forward(const AidDecoderBlock* blk, int cin, int cout, int upsample, int R, int h, const float* x, float* out, void* ws, size_t ws_bytes,
                              void* stream);

/* FunctionSpaceEpistemicEstimator.compute_jacobian_features (active_inference.py:940-993) as a call of its own: z [R,L] -> out [R, jp4.out].
 * delta [4,R,L]: standard-normal perturbation directions (normalised and scaled by perturbation_scale inside, :962; NULL => device Philox from
 * (seed, offset)); proj_mask [R, jp0.out]: jacobian_projector dropout mask in train mode (NULL => drawn).  The decoder runs in eval mode (:949). */
size_t aid_jacobian_features_workspace(const AidDecoderNet* dec, int R);
int aid_jacobian_features(const AidDecoderNet* dec, const AidEpistemicNet* epi, int R, const float* z, const float* delta, const float* proj_mask, int train,
                          uint64_t seed, uint64_t offset, float* out, void* ws, size_t ws_bytes, void* stream);

/* Pixel-mode epistemic value for ONE horizon step of all K trajectories (gradient-free, active_inference.py:398-418 + the pixel estimator):
 * next_mean [K*B, L]; noise: mine_z_eps [K*S*B, L], mine_delta [4, K*S*B, L], proj_mask [K*S*B, 512], mine_mask [4, K*S*B, 512], perms [K*S, B];
 * attn_mask [4][K*S*B*8][16][441] scaled keep masks of the aggregator's attention dropout (NULL members => device Philox).
 * mine_out [K][4] = {joint, marginal term, t_exp, mi} per trajectory.  ~40 GFLOP per sample row: callers keep K*S*B small. */
size_t aid_pixel_epistemic_workspace(const AidConvDecoder* dec, int K, int S, int B);
int aid_pixel_epistemic_value(const AidConvDecoder* dec, const AidPixelEpistemicNet* epi, int K, int S, int B, const float* next_mean,
                              const AidEfeNoise* noise, const float* attn_mask, int train, uint64_t seed, uint64_t offset, float* mine_out,
                              void* ws, size_t ws_bytes, void* stream);
/* The pixel branch of compute_jacobian_features (:976-983): ConvDecoder finite differences -> pixel_processor -> spatial_aggregator -> jacobian_projector.
 * attn_mask: the aggregator's attention-dropout mask in train mode (NULL => drawn). */
size_t aid_pixel_jacobian_features_workspace(const AidConvDecoder* dec, int R);
int aid_pixel_jacobian_features(const AidConvDecoder* dec, const AidPixelEpistemicNet* epi, int R, const float* z, const float* delta, const float* proj_mask,
                                const float* attn_mask, int train, uint64_t seed, uint64_t offset, float* out, void* ws, size_t ws_bytes, void* stream);
/* Adds the epistemic term to an EFE computed with skip_epistemic: efe[b] += sum_t discount^t * w * mean_k max(mi[t][k], 0); info[1,4,5,6]
 * (epistemic_mean and the last (k,t)'s mi / joint / marginal) and the running-mean EMA replayed in the reference's (k outer, t inner) order. */
int aid_efe_add_epistemic(const float* mine_all /*[H][K][4]*/, int H, int K, int B, float epistemic_weight, float discount, float alpha,
                          float* running_mean, float* efe, float* info, void* stream);
size_t aid_efe_step_workspace(const AidPolicyNet* pol, const AidDecoderNet* dec, int B, const AidEfeCfg* cfg);
int aid_efe_step(const AidPolicyNet* pol, const AidPolicyNet* pgrads, const AidDynamicsNet* dyn, const AidRewardNet* rew,
                 const AidValueNet* val, const AidDecoderNet* dec, const AidEpistemicNet* epi, int B, const float* latents,
                 const AidEfeNoise* noise, const AidEfeCfg* cfg, float* running_mean, float* efe_out, float* info,
                 void* ws, size_t ws_bytes, void* stream);

/* sum of squares of a gradient arena -> scratch[2] (and its sqrt in scratch[0]); aid_adamw_step_ext adds *ext_sumsq to its own
 * sum before clipping, so that several arenas are clipped by ONE global norm (pixel mode: score+diffusion and encoder arenas,
 * aid/agents/pixel_agent.py:326-331). */
int aid_grad_sumsq(const float* g, size_t n, float* scratch, void* stream);
int aid_adamw_step_ext(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float weight_decay, float beta1,
                       float beta2, float eps, int step, float clip_norm, float ema_decay, float* scratch, const float* ext_sumsq,
                       void* stream);
/* torch.optim.Adam with weight_decay (L2 folded into the gradient, NOT decoupled) after clip_grad_norm_: the estimator's optimiser,
 * base_agent.py:134-139. */
int aid_adam_l2_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float weight_decay, float beta1, float beta2, float eps,
                     int step, float clip_norm, float* scratch, void* stream);

/* ---- pixel mode (aid/agents/pixel_agent.py:273-437, aid/core/active_inference.py:142-149) --------------------------------------
 * feature decoder Linear(L,D) LN ReLU Dropout(.2) Linear(D,L) Tanh: recon = mse(fd(latents), target); loss contribution +recon.
 * Accumulates parameter gradients into fgrads and d(recon)/d(target) into d_target [B,L]; recon -> *loss_out (device). */
typedef struct AidFeatureDecoder { AidLinear l0; AidLN ln; AidLinear l4; } AidFeatureDecoder;
/* forward only: out [B, L] = tanh(Linear(Dropout(ReLU(LN(Linear(latents))))))  (decode_observation(decode_to_pixels=False), :227-231) */
int aid_feature_decoder_forward(const AidFeatureDecoder* fd, int B, const float* latents, const float* drop_mask, float* out, void* ws, size_t ws_bytes,
                                void* stream);
int aid_feature_decoder_step(const AidFeatureDecoder* fd, const AidFeatureDecoder* fgrads, int B, const float* latents,
                             const float* target, const float* drop_mask, float* d_target, float* loss_out, void* ws, size_t ws_bytes,
                             void* stream);
/* InfoNCE representation loss (pixel_agent.py:413-437): pred = predict_next_latent(latents, actions); logits =
 * normalize(pred) normalize(next_enc)^T / 0.1 * 1/(1+mean std); CE against the diagonal.  loss -> *loss_out;
 * weight * d(loss)/d(next_enc) is accumulated into d_next_enc [B,L].  (Dynamics gradients are discarded by the reference.) */
int aid_infonce_step(const AidDynamicsNet* dyn, int B, const float* latents, const float* actions, const float* next_enc, float weight,
                     float* d_next_enc, float* loss_out, void* ws, size_t ws_bytes, void* stream);
/* RandomShiftAugmentation as a stand-alone op: out[b,c,h,w] = reflect-padded x cropped at (shift_h[b], shift_w[b]); NCHW in/out. */
int aid_random_shift(const float* x, const int32_t* shift_h, const int32_t* shift_w, int pad, int B, int C, int H, int W, float* out,
                     void* stream);

/* ---- A11 + value/dynamics optimiser phases (state_agent.py:183-241) -----------------------------
 * lambda-returns (active_inference.py:638-707; minibatch order treated as time), Huber value loss with
 * gradients, MSE dynamics loss with gradients.  dones: [B] floats (0/1).  loss_out: device scalar. */
int aid_lambda_returns(const float* rewards, const float* next_values, const float* dones, int B, float gamma, float lam,
                       int n_steps, float* out, void* stream);
/* exclude_immediate_rewards != 0: the k = 0 reward of every n-step return is left out (active_inference.py:649-653, 674-675, 701-702).
 * `lam` is a double here and in aid_value_step[_ex]: the reference forms the weights (1 - lambda) lambda^i from a Python float (fp64) and only
 * then rounds them to float32; a float 0.95f already differs in the weights' last bit (aid_lambda_returns keeps its float argument). */
int aid_lambda_returns_ex(const float* rewards, const float* next_values, const float* dones, int B, float gamma, double lam,
                          int n_steps, int exclude_immediate_rewards, float* out, void* stream);
size_t aid_value_step_workspace(const AidValueNet* val, int B);
int aid_value_step(const AidValueNet* val, const AidValueNet* vgrads, int B, const float* latents, const float* next_latents,
                   const float* rewards, const float* dones, float gamma, double lam, int n_steps, float* targets_out,
                   float* loss_out, void* ws, size_t ws_bytes, void* stream);
/* aid_value_step with a lambda-return HALO: next_latents / rewards / dones hold B_ext >= B rows; the first B are this call's minibatch,
 * rows B..B_ext-1 continue the sequence (data parallel: the first n_steps rows of the next rank's shard), so that the windows of the
 * last rows see the same neighbours as in one global batch (compute_lambda_returns treats minibatch order as time, :638-707).
 * Targets, Huber loss (mean over B) and gradients are those of the first B rows. */
int aid_value_step_ex(const AidValueNet* val, const AidValueNet* vgrads, int B, int B_ext, const float* latents, const float* next_latents,
                      const float* rewards, const float* dones, float gamma, double lam, int n_steps, float* targets_out,
                      float* loss_out, void* ws, size_t ws_bytes, void* stream);
size_t aid_dynamics_step_workspace(const AidDynamicsNet* dyn, int B);
int aid_dynamics_step(const AidDynamicsNet* dyn, const AidDynamicsNet* dgrads, int B, const float* latents, const float* actions,
                      const float* next_latents, float* loss_out, void* ws, size_t ws_bytes, void* stream);

/* forward-only helpers used by act() and tests */
size_t aid_small_net_workspace(int B, int D);
int aid_policy_forward(const AidPolicyNet* pol, int B, const float* z, const float* eps /*NULL = deterministic*/, float* action,
                       float* log_prob, float* entropy, void* ws, size_t ws_bytes, void* stream);
int aid_value_forward(const AidValueNet* val, int B, const float* state, const float* t, float* out, void* ws, size_t ws_bytes,
                      void* stream);
int aid_predict_next_latent(const AidDynamicsNet* dyn, int B, const float* z, const float* a, float* next_mean, void* ws,
                            size_t ws_bytes, void* stream);
int aid_reward_forward(const AidRewardNet* rew, int B, const float* z, float* out2, void* ws, size_t ws_bytes, void* stream);

/* A15 FreeEnergyComputation.compute_loss reductions (aid/core/free_energy.py:62-78); obs must be [B,L] like states.
 * out3 = mean_b sum_l of (z-mu_p)^2, (o-z)^2, score^2. */
int aid_free_energy_terms(const float* states, const float* prior_mean, const float* obs, const float* score, int B, int L,
                          float* out3, void* stream);
/* ---- A5: LatentDiffusionProcess's forward-process helpers as stand-alone operations (aid/core/diffusion.py).  One launch each; inside a
 * train_step the same arithmetic runs fused in aid_elbo_step.  log_snr_min / log_snr_max: device scalars (the module's Parameters).
 *   aid_log_snr               compute_log_snr        :56-60    out[i] = min + (max - min) (1 - t[i]); n elements of any shape
 *   aid_loss_weight           compute_loss_weight    :93-104   out[i] = exp(-lambda^2 / 8) (sin(pi t) + 0.1)
 *   aid_continuous_q_sample   continuous_q_sample    :62-91    z_noisy [B,L] = sqrt(alpha_b) z + sqrt(sigma_b) eps; log_snr / alpha / sigma [B] optional (NULL)
 *   aid_sample_latent_prior   sample_latent_prior    :146-152  out [B,L] = prior_mean + exp(prior_log_std) eps
 *   aid_q_sample              q_sample               :154-174  per-row int64 steps in [0, N); tables [N] = the module's sqrt_alphas_cumprod buffers
 *   aid_posterior_mean        _posterior_mean        :239-255  coef1 = beta sqrt(acp_prev) / (1 - acp), coef2 = (1 - acp_prev) sqrt(alpha) / (1 - acp), [N] each
 * bad_step (optional device int, zeroed by the caller): set to 1 when a step lies outside [0, N) - the reference's gather raises there; the kernel
 * itself clamps, it never reads outside the tables. */
int aid_log_snr(const float* t, const float* log_snr_min, const float* log_snr_max, float* out, size_t n, void* stream);
int aid_loss_weight(const float* t, const float* log_snr_min, const float* log_snr_max, float* out, size_t n, void* stream);
int aid_continuous_q_sample(const float* z_start, const float* t, const float* eps, const float* log_snr_min, const float* log_snr_max, int B, int L,
                            float* z_noisy, float* log_snr, float* alpha, float* sigma, void* stream);
int aid_sample_latent_prior(const float* prior_mean, const float* prior_log_std, const float* eps, int B, int L, float* out, void* stream);
int aid_q_sample(const float* z_start, const int64_t* steps, const float* eps, const float* sqrt_alphas_cumprod, const float* sqrt_one_minus_alphas_cumprod,
                 int N, int B, int L, float* z_noisy, int* bad_step, void* stream);
int aid_posterior_mean(const float* z_start, const float* z_t, const int64_t* steps, const float* coef1, const float* coef2, int N, int B, int L, float* out,
                       int* bad_step, void* stream);
/* p_sample with a caller-supplied score (aid/core/diffusion.py:208-255); eps may be NULL (deterministic / t = 0). */
int aid_p_sample(const float* z, const float* score, const float* eps, const float* coefs, int step, int B, int L, float* out,
                 void* stream);
/* A0': RunningMeanStd.update + normalize (aid/agents/base_agent.py:24-52), fp64 on device.  state: [3] doubles
 * (mean, var, count), updated in place; out[B] = (r - mean)/sqrt(var + 1e-8) with the UPDATED statistics. */
int aid_reward_normalize(const float* rewards, int B, double* state, float* out, void* stream);
/* data-parallel form: statistics are updated from `stats_src` (the rewards of ALL ranks, all-gathered, n_stats of
 * them), `rewards` (this rank's B) are normalised with the updated statistics. */
int aid_reward_normalize_global(const float* stats_src, int n_stats, const float* rewards, int B, double* state, float* out,
                                void* stream);


/* ---- A13: DrQ-v2 pixel encoder (aid/encoder/visual_encoders.py:13-225) + RandomShiftAugmentation (:471-492) ----------
 * 4 x {conv3x3 (no bias, legacy spectral norm; stride 2,1,1,1; in_ch->32->64->128->256) -> GroupNorm -> Mish -> Dropout2d}
 * -> SpatialAttention -> flatten (NCHW order) -> LayerNorm(256*42*42) -> Linear -> LN -> Mish -> Dropout(.1) -> Linear -> LN -> tanh.
 * x: [B, in_ch, H, W] float32 (frame-stacked, in [0,1]).  shift_h / shift_w: per-sample crop offsets in [0, 2*pad] of the
 * reflect-padded image (NULL = no augmentation).  drop2d_1 [B,64], drop2d_2 [B,128], drop_out [B, 2*feature_dim]: scaled keep masks
 * (NULL in eval mode).  train=1 runs the power iteration and UPDATES conv_u / conv_v in place (as the reference's buffers).
 * `saved` (aid_encoder_saved_bytes) carries the activations to aid_encoder_backward, which ACCUMULATES parameter gradients
 * into `grads` (conv_w = d weight_orig; conv_u/conv_v ignored). */
typedef struct AidEncoderNet {
    int32_t feature_dim, in_ch, H, W;
    float *conv_w[4], *conv_u[4], *conv_v[4];            /* convs.i.weight_orig / weight_u / weight_v */
    float *gn_g[4], *gn_b[4];                            /* norms.i.weight / bias */
    float *att_temperature, *att_conv_w, *att_conv_b;    /* attention.temperature / spatial_conv.{weight,bias} */
    float *ln_g, *ln_b;                                  /* ln.weight / bias [256*42*42] */
    AidLinear out0; AidLN out1; AidLinear out4; AidLN out5;   /* output_layers.{0,1,4,5} */
} AidEncoderNet;
size_t aid_encoder_saved_bytes(const AidEncoderNet* enc, int B);
size_t aid_encoder_workspace(const AidEncoderNet* enc, int B);
int aid_encoder_forward(const AidEncoderNet* enc, int B, const float* x, const int32_t* shift_h, const int32_t* shift_w, int pad,
                        const float* drop2d_1, const float* drop2d_2, const float* drop_out, int train, float* features, void* saved,
                        size_t saved_bytes, void* ws, size_t ws_bytes, void* stream);
int aid_encoder_backward(const AidEncoderNet* enc, const AidEncoderNet* grads, int B, const float* d_features, const float* drop2d_1,
                         const float* drop2d_2, const float* drop_out, int train, const void* saved, size_t saved_bytes, void* ws,
                         size_t ws_bytes, void* stream);
/* The same backward in two parts, for a data-parallel caller that wants the all-reduce of the Linear(451 584 -> 2L) gradient (462 MB at L = 128: the first thing the
 * backward produces) under way while the conv stack is still being differentiated.  phase 1: output layers, that Linear and LayerNorm(F) - writes the
 * gradients of ln.* / output_layers.* and leaves d(loss)/d(attention output) in `ws`; phase 2: spatial attention + conv stack, reading it (d_features may be
 * NULL).  Both phases of one pass take the SAME `ws`, untouched in between; phase 0 = aid_encoder_backward. */
int aid_encoder_backward_phase(const AidEncoderNet* enc, const AidEncoderNet* grads, int B, const float* d_features, const float* drop2d_1,
                               const float* drop2d_2, const float* drop_out, int train, const void* saved, size_t saved_bytes, void* ws,
                               size_t ws_bytes, int phase, void* stream);

/* ---- A0: ReplayBuffer.sample gather (aid/utils/buffers.py:83-111), device-resident ring arrays.
 * idx: [B] int64 row indices.  One coalesced pass over the 2*O + A + 2 floats of each transition. */
int aid_replay_gather(const float* obs, const float* next_obs, const float* actions, const float* rewards, const float* dones,
                      const int64_t* idx, int B, int O, int A, float* obs_out, float* next_obs_out, float* act_out,
                      float* rew_out, float* done_out, void* stream);

/* Same with uint8 frame storage (obs / next_obs rows are bytes; outputs are float32 with the same values 0..255). */
int aid_replay_gather_u8(const uint8_t* obs, const uint8_t* next_obs, const float* actions, const float* rewards, const float* dones,
                         const int64_t* idx, int B, int O, int A, float* obs_out, float* next_obs_out, float* act_out,
                         float* rew_out, float* done_out, void* stream);

/* ---- A14: Fokker-Planck belief step, diagonal covariance, fp64 (aid/core/belief_dynamics.py:97-172) ---
 * Batched over n beliefs of dimension L (double pointers): mean/var updated in place; entropy_out [n] optional.
 * xi: [n,L] N(0,1) draws (double).  Closed form documented in oracle.aid_oracle.fokker_planck_step. */
int aid_fokker_planck_step(double* mean, double* var, const double* obs, const double* score, const double* xi, int n, int L,
                           double dt, double D, double lr, double noise_scale, double min_var, double max_var,
                           double* entropy_out, void* stream);

/* ---- A14, full covariance (aid/core/belief_dynamics.py:141-152, 268-342, 368-383) ---------------------
 * With the Gaussian observation model the Hessian is a multiple of I, so the matrix exponential is a scalar and the eigen-clamp,
 * condition-number regularisation, (cov + eps I)^-1 and logdet are spectral functions of the covariance: the belief is carried as
 * eigenvalues lam [n,L] + eigenvectors V [n,L,L] (columns), decomposed once at reset.
 * aid_sym_eig_f64: symmetric eigen-decomposition (parallel cyclic Jacobi, fp64, one workgroup): A [n,n] is destroyed, w [n] unsorted,
 *   V [n,n] eigenvectors in columns (replaces torch.linalg.eigh of `_ensure_numerical_stability` :303).
 * aid_fokker_planck_full_step: advance=1: mean step as the diagonal entry point, lam *= exp(2 (-2h + 2D) dt) (`_matrix_exponential_update`);
 *   advance=0: the stabilisation of reset(initial_cov) only.  Then lam = max(lam, eps), + mean(lam) 1e-6 if max/min > 1e6, and
 *   cov_out = V diag(lam) V^T, prec_out = V diag(1/(lam+eps)) V^T  [n,L,L]; entropy_out / cond_out [n] optional.  L <= 128. */
int aid_sym_eig_f64(double* A, double* w, double* V, int n, void* stream);
int aid_fokker_planck_full_step(double* mean, double* lam, const double* V, const double* obs, const double* score, const double* xi, int n, int L,
                                double dt, double D, double lr, double noise_scale, double min_var, int advance, double* cov_out,
                                double* prec_out, double* entropy_out, double* cond_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AID_HIP_H */
