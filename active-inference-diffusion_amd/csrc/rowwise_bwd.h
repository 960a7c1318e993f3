// Backward / tangent row-wise kernels (derivations: oracle/derivations/score_manual_backward.py).
#pragma once
#include "aid_common.h"

namespace aid {

// out[c] (+)= sum_r a[r*lda + c]                       (bias gradients; deterministic)
int colsum(const float* a, int lda, float* out, int M, int N, int accumulate, hipStream_t s, Workspace* ws = nullptr);

// LayerNorm(+affine)(+activation)(+dropout mask) backward, rows:  dx = T(dyeff*gamma), dyeff = dy*mask*act'(xhat*gamma+beta)
int ln_bwd_rows(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                int act, const float* mask, float* dx, int lddx, float* dyeff, int M, int D, hipStream_t s);
// dgamma[c] += sum_r dyeff*xhat ; dbeta[c] += sum_r dyeff
int ln_param_grad(const float* dyeff, const float* x, int ldx, const float* stats, float* dgamma, float* dbeta, int M, int D, hipStream_t s,
                  Workspace* ws = nullptr);

// adaLN backward carrying the primal adjoint (mu) and, optionally, the tangent adjoint (lam):
//   y = xhat*(1+scale)+shift ;  yd = T(h)[hd]*(1+scale)
// mu_out = mu_in + T(mu_nx*(1+s)) + d/dh[lam_a . T(h)[hd]] ; lam_out = lam_in + T(lam_a), lam_a = lam_nxd*(1+s)
// dmod[row, c] = mu_nx*xhat + lam_nxd*T(h)[hd] ; dmod[row, D+c] = mu_nx.     Null pointers drop their terms.
int adaln_bwd_dual(const float* h, int ldh, const float* stats, const float* mod, int ldmod, const float* mu_nx,
                   const float* hd, const float* lam_nxd, const float* mu_in, const float* lam_in, float* mu_out,
                   float* lam_out, float* dmod, int lddmod, int M, int D, hipStream_t s);
// tangent of adaLN: nxd = T(h)[hd]*(1+scale)
int adaln_tangent(const float* h, int ldh, const float* stats, const float* mod, int ldmod, const float* hd, float* nxd, int M,
                  int D, hipStream_t s);
// through y = act(pre): mu = mu*act'(pre) + lam*xd*act''(pre) ; lam = lam*act'(pre)   (lam/xd may be null)
int act_bwd_dual(const float* pre, float* mu, float* lam, const float* xd, int act, size_t n, hipStream_t s);

// score head backward (score = clamp(raw,-10,10)*mult*tw):
//   rawd == null: mu_raw = d_score*mult*tw*inside (d_score null -> ones); dmult += sum d_score*clamp(raw)*tw
//   rawd != null: lam_raw = coef*inside*mult*tw, mu_raw = 0;              dmult += coef*sum inside*tw*rawd
int head_bwd(const float* raw, const float* d_score, const float* mult, const float* t, int mode, const float* rawd, float coef,
             float* mu_raw, float* lam_raw, float* dmult, int M, int L, hipStream_t s);

int sinusoid_bwd(const float* dsin, const float* t, const float* fs, float tmul, int mode, float* dfs, int M, int dim, hipStream_t s);
int dot_accum(const float* a, const float* b, size_t n, float scale, float* out, hipStream_t s);     // out[0] += scale*sum a*b
int scale_by_ptr(const float* a, const float* scalar, float* out, size_t n, hipStream_t s);           // out = scalar[0]*a
int add_inplace(float* a, const float* b, size_t n, hipStream_t s);                                   // a += b

// gradient-penalty weights: pen = mean_b (||g_b||-1)^2 ; w_b = (2/B)(||g_b||-1) g_b/||g_b|| (0 where the norm is 0)
int gp_weights(const float* g, float* w, float* pen, int B, int L, hipStream_t s);

}  // namespace aid
