// MLP building blocks with saved activations (forward + hand-written backward) and the small networks
// of the hot path: policy, value, dynamics, reward predictor, observation decoder.
#pragma once
#include "aid_common.h"
#include "rowwise.h"
#include "rowwise_bwd.h"

namespace aid {

int copy_cols(const float* src, int lds, float* dst, int ldd, int M, int nc, hipStream_t s);   // lds==0 broadcasts a row
int axpby_cols(const float* a, int lda, float alpha, const float* b, int ldb, float beta, float* dst, int ldd, int M, int nc, hipStream_t s);

// Linear -> LayerNorm -> activation (-> + add) in ONE full-row-tile MFMA kernel (gemm_rowln.hip): tall inference inputs, N in {512, 1024}
bool linear_ln_act_rows_ok(int M, int N, int K, int act, const float* x, int ldx, const float* W, const float* add, int ldadd, const float* y, int ldy);
int linear_ln_act_rows(const float* x, int ldx, const float* W, const float* b, const float* gamma, const float* beta, int act, const float* add,
                       int ldadd, float* y, int ldy, int M, int N, int K, hipStream_t s, const float* mask = nullptr, int ldmask = 0);

// y = act(LN(x W^T + b)) * mask      (Linear -> LayerNorm -> activation [-> dropout])
struct LlaSave { float *lin, *stats, *y; };
int lla_alloc(int M, int out, Workspace& ws, LlaSave& sv);
int lla_fwd(const AidLinear& l, const AidLN& ln, int act, const float* mask, const float* x, int ldx, int M, LlaSave& sv, hipStream_t s,
            const float* add = nullptr, bool need_lin = true);
// dy: gradient wrt sv.y [M,out].  Writes dx (optional, row stride lddx) ; accumulates parameter grads when gl/gln given.
// dlin_keep (optional, [M,out]): the adjoint of the Linear output is left there and the weight / bias gradients are NOT computed: the caller
// contracts them later (GradQueue: one grouped launch per backward pass, gemm_group.hip).
int lla_bwd(const AidLinear& l, const AidLN& ln, const AidLinear* gl, const AidLN* gln, int act, const float* mask, const float* x,
            int ldx, int M, const LlaSave& sv, const float* dy, float* dx, int lddx, Workspace& ws, hipStream_t s, float* dlin_keep = nullptr,
            float* dyeff_keep = nullptr);      // dyeff_keep: the operand of the LayerNorm gamma / beta gradients is left there, their column sums to the caller

// y = act(x W^T + b) * mask            (Linear -> activation), pre-activation saved
struct LaSave { float *pre, *y; };
int la_alloc(int M, int out, bool need_pre, Workspace& ws, LaSave& sv);
int la_fwd(const AidLinear& l, int act, const float* mask, const float* x, int ldx, int M, LaSave& sv, int ldy, hipStream_t s);
// dy is overwritten (multiplied by act'(pre)*mask).
// defer_params: dy (which then holds the adjoint of the Linear output) is the operand of a gradient the caller contracts later.
int la_bwd(const AidLinear& l, const AidLinear* gl, int act, const float* mask, const float* x, int ldx, int M, const LaSave& sv,
           float* dy, int lddy, float* dx, int lddx, hipStream_t s, int accumulate_dx = 0, Workspace* ws = nullptr, bool defer_params = false);

// ---- observation decoder (state branch), aid/core/active_inference.py:237-242
struct DecoderSave { LlaSave s0, s1, s2; float* h2; };
size_t decoder_ws_bytes(const AidDecoderNet& d, int M);
int decoder_fwd(const AidDecoderNet& d, int M, const float* z, int ldz, const float* const masks[3], float* out, int ldo, Workspace& ws,
                hipStream_t s);

// ---- reward predictor, aid/core/active_inference.py:164-171
struct RewardSave { LlaSave s0; LaSave s1; float* out; };
int reward_fwd(const AidRewardNet& r, int M, const float* z, int ldz, RewardSave& sv, Workspace& ws, hipStream_t s);
// d_out: [M,2] gradient wrt the output (overwritten); dz accumulates? -> writes dz [M,L] (row stride lddz)
int reward_bwd_data(const AidRewardNet& r, int M, const float* z, int ldz, const RewardSave& sv, float* d_out, float* dz, int lddz,
                    Workspace& ws, hipStream_t s);

// ---- dynamics, aid/models/dynamics_models.py:47-68 (+ the second residual of active_inference.py:461-462)
struct DynSave { float* x; LlaSave s[3]; float* out; };   // x = [z | a]
int dynamics_fwd(const AidDynamicsNet& d, int M, const float* z, int ldz, const float* a, int lda, float* next_mean, DynSave& sv,
                 Workspace& ws, hipStream_t s);
// d_next: gradient wrt next_mean = 2z + net([z|a]).  dz_out = 2 d_next + dx[:, :L] ; da_out = dx[:, L:]
int dynamics_bwd(const AidDynamicsNet& d, const AidDynamicsNet* g, int M, const DynSave& sv, const float* d_next, float* dz_out,
                 float* da_out, Workspace& ws, hipStream_t s, GradQueue* q = nullptr);      // q: weight / bias gradients recorded, the caller flushes

// ---- value network, aid/models/value_networks.py:47-60
struct ValueSave { float *sin_e, *te_pre, *x; LlaSave s[3]; float* out; };   // x = [state | relu(te1(sinus(t)))]
int value_fwd(const AidValueNet& v, int M, const float* state, int lds, const float* t, float* out, ValueSave& sv, Workspace& ws, hipStream_t s);
// d_out [M] gradient wrt V.  dstate (optional) [M,L].  Parameter grads when g != null.
int value_bwd(const AidValueNet& v, const AidValueNet* g, int M, const float* t, const ValueSave& sv, const float* d_out, float* dstate,
              int ldds, Workspace& ws, hipStream_t s, GradQueue* q = nullptr);

// ---- policy, aid/models/policy_networks.py:95-146
struct PolicySave { LlaSave e0; float* h; LlaSave tr[3]; float* h2; LaSave m0, s0; float *mean, *lstd, *action, *ent, *logp; bool prealloc = false; };
int policy_fwd(const AidPolicyNet& p, int M, const float* z, int ldz, const float* eps, PolicySave& sv, Workspace& ws, hipStream_t s);
// The H forward passes of a rollout saved side by side: every saved tensor (and every adjoint a weight gradient reads) of step t is slice t of
// ONE [H][M][width] buffer, so that the BPTT's parameter gradients dW = sum_t adj_t^T x_t are single contractions over H*M rows
// (policy_bptt_grads) instead of H launches per layer.
struct PolicySteps {
    int H, M;
    PolicySave sv[16];
    float *dmean, *dlstd, *dm0, *ds0, *dtr[3], *de0, *dh;        // [H][M][.] adjoints of mean2 / lstd2 / mean0 / lstd0 / trunk i / enc0 outputs, enc3 output
    float* dyeff[4];                                             // [H][M][D] dy act'(.) of the enc0 / trunk i LayerNorms: what their gamma / beta gradients sum
};
int policy_steps_alloc(const AidPolicyNet& p, int M, int H, Workspace& ws, PolicySteps& ps);
// record dW / db of every policy Linear over all H steps into q (z_all: [H][M][L] inputs of the H forward passes)
// (and the LayerNorm gamma / beta gradients, likewise over all H*M rows at once)
int policy_bptt_grads(const AidPolicyNet& p, const AidPolicyNet& g, const PolicySteps& ps, const float* z_all, int ldz, GradQueue& q, Workspace& ws);
// d_action [M,A] (may be null = 0), ent_coef[M] = dLoss/d(entropy_sum) per row (may be null), accumulates param grads; dz optional
// steps / t (optional): write step t's adjoints into the PolicySteps slices and leave the Linear weight / bias gradients to policy_bptt_grads
// and the LayerNorm parameter gradients to policy_bptt_grads.
int policy_bwd(const AidPolicyNet& p, const AidPolicyNet* g, int M, const float* z, int ldz, const float* eps, const PolicySave& sv,
               const float* d_action, const float* ent_coef, float* dz, int lddz, Workspace& ws, hipStream_t s, const PolicySteps* steps = nullptr, int t = 0);

// ---- implicit-GEMM 3x3 convolution on CNHW activations (conv_igemm.hip): no im2col matrix
// out[co][n = (b, ho, wo)] = sum_k W[co][k] x[ci][b][ho s + kh - 1][wo s + kw - 1], k = ci 9 + kh 3 + kw; W [Cout][ldw], ldw >= 9 Cin, ldw % 4 == 0
int conv3_igemm(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, hipStream_t s);
// dW[co][k] (row stride ldw) = sum_n dy[co][n] X(k, n); slab: scratch [S][Cout][ldw] (S row chunks folded in order)
int conv3_igemm_dw(const float* dy, const float* x, float* dW, int ldw, int Cin, int Cout, int B, int Hi, int Ho, int stride, float* slab, int S, hipStream_t s);
// Wt[ci][co 9 + kh 3 + kw] = W[co][ci 9 + (2 - kh) 3 + (2 - kw)]: the data gradient of a stride-1 conv is conv3_igemm(Wt, dy) with Cin <-> Cout
int conv3_flip_weights(const float* W, int ldw, float* Wt, int ldwt, int Cin, int Cout, hipStream_t s, int tap_major = 0);
// tap-major contraction order k' = (kh 3 + kw) Cin + ci (conv_igemm.hip, TAPM): needs Cin % 16 == 0 (% 32 when Cout <= 32)
bool conv3_tapmajor_ok(int Cin, int Cout);
int conv3_tapmajor_weights(const float* W, int ldw, float* Wtm, int ldwt, int Cin, int Cout, hipStream_t s);                      // Wtm[co][tap Cin + ci] = W[co][ci 9 + tap]
int conv3_igemm_ex(const float* W, int ldw, const float* x, float* out, int Cin, int Cout, int B, int Hi, int Ho, int stride, int w_tap_major, hipStream_t s);

// ---- ConvDecoder + pixel estimator feature side (conv.hip)
int conv_decoder_fwd(const AidConvDecoder& d, int R, const float* z, float* out, Workspace& ws, hipStream_t s);
int pixel_jacobian_features(const AidConvDecoder& dec, const AidPixelEpistemicNet& e, int R, const float* z_all, const float* delta,
                            const float* attn_mask, int train, uint64_t seed, uint64_t offset, float* jin, Workspace& ws, hipStream_t s);
size_t pixel_jacobian_ws_bytes(const AidConvDecoder& dec);

}  // namespace aid
