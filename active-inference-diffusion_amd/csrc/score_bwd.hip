// LatentScoreNetwork backward on gfx950: hand-written reverse pass (no autograd), including the
// second-order pass the gradient penalty needs (aid/core/active_inference.py:709-729).
//
// The penalty P = mean_b(||g_b||-1)^2 with g = d(sum s)/dz_t is differentiated as reverse-mode through a
// forward-mode pass: dP/dtheta = d/dtheta sum(J(theta) w), w = dP/dg held fixed.  `score_tangent_forward`
// pushes w through the saved primal activations; `score_reverse` then walks the network backwards carrying
// TWO adjoint streams: lam (adjoints of tangent activations) and mu (adjoints of primal activations, fed by
// act'' and the LayerNorm second derivative).  With no tangent it degenerates to the ordinary backward.
// Formulas are validated against autograd in oracle/derivations/score_manual_backward.py.
#include "score_net.h"
#include "rowwise_bwd.h"
#include "score_bwd.h"

namespace aid {

static inline int n_ada(const AidScoreNet& n) { return 2 * n.n_blocks + 1; }

size_t score_tan_bytes(const AidScoreNet& n, int B) {
    Workspace ws(nullptr, 0); ScoreTan t; score_tan_alloc(n, B, ws, t); return ws.off + 256;
}
int score_tan_alloc(const AidScoreNet& n, int B, Workspace& ws, ScoreTan& t) {
    const size_t D = n.D, L = n.L;
    for (int i = 0; i <= n.n_blocks; ++i) t.hd[i] = ws.f32(B * D);
    for (int i = 0; i < n.n_blocks; ++i) {
        ScoreTanBlock& b = t.blk[i];
        b.nx1d = ws.f32(B * D); b.vd = ws.f32(B * D); b.hmd = ws.f32(B * D); b.nx2d = ws.f32(B * D);
        b.a1d = ws.f32(B * 4 * D); b.ud = ws.f32(B * 4 * D);
    }
    t.nxfd = ws.f32(B * D); t.p0pd = ws.f32(B * (D / 2)); t.p0d = ws.f32(B * (D / 2)); t.rawd = ws.f32(B * L);
    return AID_OK;
}

// tangent forward along z_dot through saved activations `a` (keep=true forward).  No biases, cond has no tangent.
int score_tangent_forward(const AidScoreNet& n, int B, const float* z_dot, const ScoreActs& a, ScoreTan& t, hipStream_t s) {
    const int D = n.D, L = n.L, ldmod = n_ada(n) * 2 * D;
    AID_TRY(linear_fwd(z_dot, L, n.lp_w, nullptr, t.hd[0], D, B, D, L, ACT_NONE, s));
    for (int i = 0; i < n.n_blocks; ++i) {
        const AidDiTBlock& w = n.blk[i]; const ScoreBlockActs& b = a.blk[i]; ScoreTanBlock& tb = t.blk[i];
        AID_TRY(adaln_tangent(b.h_in, D, b.st1, a.mod + (size_t)(2 * i) * 2 * D, ldmod, t.hd[i], tb.nx1d, B, D, s));
        AID_TRY(linear_fwd(tb.nx1d, D, w.in_proj_w + (size_t)2 * D * D, nullptr, tb.vd, D, B, D, D, ACT_NONE, s));
        AID_TRY(linear_fwd(tb.vd, D, w.out_proj_w, nullptr, tb.hmd, D, B, D, D, ACT_NONE, s, nullptr, t.hd[i], D));
        AID_TRY(adaln_tangent(b.h_mid, D, b.st2, a.mod + (size_t)(2 * i + 1) * 2 * D, ldmod, tb.hmd, tb.nx2d, B, D, s));
        {   // ud = gelu'(a1) * (W1 nx2d), a1d saved
            GemmArgs g = gemm_args(tb.nx2d, D, w.mlp0_w, D, tb.ud, 4 * D, B, 4 * D, D);
            g.preact = tb.a1d; g.dact_pre = b.a1; g.dact = ACT_GELU;
            AID_TRY(gemm_f32(g, s));
        }
        AID_TRY(linear_fwd(tb.ud, 4 * D, w.mlp2_w, nullptr, t.hd[i + 1], D, B, D, 4 * D, ACT_NONE, s, nullptr, tb.hmd, D));
    }
    AID_TRY(adaln_tangent(a.h_last, D, a.st_f, a.mod + (size_t)(2 * n.n_blocks) * 2 * D, ldmod, t.hd[n.n_blocks], t.nxfd, B, D, s));
    {
        GemmArgs g = gemm_args(t.nxfd, D, n.op0_w, D, t.p0d, D / 2, B, D / 2, D);
        g.preact = t.p0pd; g.dact_pre = a.p0_pre; g.dact = ACT_SILU;
        AID_TRY(gemm_f32(g, s));
    }
    AID_TRY(linear_fwd(t.p0d, D / 2, n.op2_w, nullptr, t.rawd, L, B, L, D / 2, ACT_NONE, s));
    return AID_OK;
}

// one Linear layer in the reverse pass: dW += mu^T x (+ lam^T xd); db += colsum(mu); mu_in = mu W; lam_in = lam W.
// q != null: the parameter gradients are only RECORDED (gemm_group.hip: one launch per pass computes all of them); mu / lam / x / xd must then
// stay untouched until q->flush().  When the two adjoint streams are the two halves of one [2B, N] buffer (lam == mu + B N, and likewise their
// destinations) the two data-gradient products are one 2B-row GEMM.
static int lin_rev(const float* W, float* dW, float* db, int N, int K, int B, const float* x, int ldx, const float* xd,
                   const float* mu, const float* lam, float* mu_in, float* lam_in, bool params, hipStream_t s, GradQueue* q = nullptr) {
    if (params) {
        if (q) {
            AID_TRY(q->add(dW, K, N, K, mu, N, x, ldx, B, db));
            if (lam) AID_TRY(q->add(dW, K, N, K, lam, N, xd, K, B, nullptr));
        } else {
            AID_TRY(linear_bwd_weight(mu, N, x, ldx, dW, B, N, K, s, 1));
            if (lam) AID_TRY(linear_bwd_weight(lam, N, xd, K, dW, B, N, K, s, 1));
            if (db) AID_TRY(colsum(mu, N, db, B, N, 1, s));
        }
    }
    if (mu_in && lam && lam_in && lam == mu + (size_t)B * N && lam_in == mu_in + (size_t)B * K)
        return linear_bwd_data(mu, N, W, K, mu_in, K, 2 * B, N, s);
    if (mu_in) AID_TRY(linear_bwd_data(mu, N, W, K, mu_in, K, B, N, s));
    if (lam && lam_in) AID_TRY(linear_bwd_data(lam, N, W, K, lam_in, K, B, N, s));
    return AID_OK;
}

size_t score_reverse_ws_bytes(const AidScoreNet& n, int B) {
    // two adjoint streams, side by side: head buffers + 2 shared temporaries + per block (4D + 3D) that the recorded weight gradients keep alive
    const size_t D = n.D, per_stream = (size_t)B * (n.L + (D / 2) + 3 * D + (size_t)n.n_blocks * 7 * D);
    return sizeof(float) * (2 * per_stream + 64) + (size_t)(8 + 8 * n.n_blocks) * 256;
}

// q != null (with gr): weight / bias gradients are recorded in q; the caller flushes q and only then releases the workspace mark it took
// before this call (every adjoint a recorded gradient reads lives in `ws`).  q == null: gradients are launched one by one, the workspace is
// released here.
int score_reverse(const AidScoreNet& n, const AidScoreNet* gr, int B, const ScoreActs& a, const ScoreTan* t, const float* z,
                  const float* z_dot, const float* d_score, const float* tvals, int t_mode, float coef, float* dz, float* dmod,
                  Workspace& ws, hipStream_t s, GradQueue* q) {
    const int D = n.D, L = n.L, NA = n_ada(n), ldmod = NA * 2 * D;
    const bool dual = t != nullptr, params = gr != nullptr;
    const size_t mark = ws.mark();
    // every adjoint is a [2][B][width] buffer: primal adjoint mu first, tangent adjoint lam right behind it, so that mu_in = mu W and
    // lam_in = lam W are ONE 2B-row product (lin_rev).  Buffers a weight gradient reads (the adjoints of every Linear OUTPUT) are never
    // written again: with a GradQueue their gradients are contracted after the whole chain has run.
    auto pair = [&](size_t width, float*& mu, float*& lam) { mu = ws.f32((dual ? 2 : 1) * (size_t)B * width); lam = (dual && mu) ? mu + (size_t)B * width : nullptr; };
    float *mu_raw, *lam_raw, *mu_p, *lam_p, *mu_h, *lam_h, *mu_d, *lam_d, *mu_nx, *lam_nx;
    pair(L, mu_raw, lam_raw); pair(D / 2, mu_p, lam_p); pair(D, mu_h, lam_h); pair(D, mu_d, lam_d); pair(D, mu_nx, lam_nx);
    AID_WS_OK(ws);

    AID_TRY(head_bwd(a.raw, d_score, n.output_multiplier, tvals, t_mode, dual ? t->rawd : nullptr, coef, mu_raw, lam_raw,
                     params ? gr->output_multiplier : nullptr, B, L, s));
    AID_TRY(lin_rev(n.op2_w, params ? gr->op2_w : nullptr, nullptr, L, D / 2, B, a.p0, D / 2, dual ? t->p0d : nullptr, mu_raw, lam_raw,
                    mu_p, lam_p, params, s, q));
    AID_TRY(act_bwd_dual(a.p0_pre, mu_p, lam_p, dual ? t->p0pd : nullptr, ACT_SILU, (size_t)B * (D / 2), s));
    AID_TRY(lin_rev(n.op0_w, params ? gr->op0_w : nullptr, params ? gr->op0_b : nullptr, D / 2, D, B, a.nxf, D, dual ? t->nxfd : nullptr,
                    mu_p, lam_p, mu_d, lam_d, params, s, q));
    AID_TRY(adaln_bwd_dual(a.h_last, D, a.st_f, a.mod + (size_t)(2 * n.n_blocks) * 2 * D, ldmod, mu_d, dual ? t->hd[n.n_blocks] : nullptr,
                           lam_d, nullptr, nullptr, mu_h, lam_h, dmod ? dmod + (size_t)(2 * n.n_blocks) * 2 * D : nullptr, ldmod, B, D, s));
    for (int i = n.n_blocks - 1; i >= 0; --i) {
        const AidDiTBlock& w = n.blk[i]; const ScoreBlockActs& b = a.blk[i];
        const AidDiTBlock* g = params ? &gr->blk[i] : nullptr;
        const ScoreTanBlock* tb = dual ? &t->blk[i] : nullptr;
        float *mu_u, *lam_u, *mu_m, *lam_m, *mu_v, *lam_v, *mu_o, *lam_o;      // adjoints of a1 (then u), h_mid, v and the block input
        pair(4 * D, mu_u, lam_u); pair(D, mu_m, lam_m); pair(D, mu_v, lam_v); pair(D, mu_o, lam_o);
        AID_WS_OK(ws);
        AID_TRY(lin_rev(w.mlp2_w, g ? g->mlp2_w : nullptr, g ? g->mlp2_b : nullptr, D, 4 * D, B, b.u, 4 * D, tb ? tb->ud : nullptr, mu_h, lam_h,
                        mu_u, lam_u, params, s, q));
        AID_TRY(act_bwd_dual(b.a1, mu_u, lam_u, tb ? tb->a1d : nullptr, ACT_GELU, (size_t)B * 4 * D, s));
        AID_TRY(lin_rev(w.mlp0_w, g ? g->mlp0_w : nullptr, g ? g->mlp0_b : nullptr, 4 * D, D, B, b.nx2, D, tb ? tb->nx2d : nullptr, mu_u, lam_u,
                        mu_d, lam_d, params, s, q));
        AID_TRY(adaln_bwd_dual(b.h_mid, D, b.st2, a.mod + (size_t)(2 * i + 1) * 2 * D, ldmod, mu_d, tb ? tb->hmd : nullptr, lam_d, mu_h, lam_h,
                               mu_m, lam_m, dmod ? dmod + (size_t)(2 * i + 1) * 2 * D : nullptr, ldmod, B, D, s));
        // attention (single token): out_proj(V_proj(nx1)); gradient lands in rows [2D,3D) of in_proj
        AID_TRY(lin_rev(w.out_proj_w, g ? g->out_proj_w : nullptr, g ? g->out_proj_b : nullptr, D, D, B, b.v, D, tb ? tb->vd : nullptr, mu_m, lam_m,
                        mu_v, lam_v, params, s, q));
        AID_TRY(lin_rev(w.in_proj_w + (size_t)2 * D * D, g ? g->in_proj_w + (size_t)2 * D * D : nullptr, g ? g->in_proj_b + 2 * D : nullptr, D, D, B,
                        b.nx1, D, tb ? tb->nx1d : nullptr, mu_v, lam_v, mu_nx, lam_nx, params, s, q));
        AID_TRY(adaln_bwd_dual(b.h_in, D, b.st1, a.mod + (size_t)(2 * i) * 2 * D, ldmod, mu_nx, dual ? t->hd[i] : nullptr, lam_nx, mu_m, lam_m,
                               mu_o, lam_o, dmod ? dmod + (size_t)(2 * i) * 2 * D : nullptr, ldmod, B, D, s));
        mu_h = mu_o; lam_h = lam_o;
    }
    AID_TRY(lin_rev(n.lp_w, params ? gr->lp_w : nullptr, params ? gr->lp_b : nullptr, D, L, B, z, L, z_dot, mu_h, lam_h, dz, nullptr, params, s, q));
    if (!q) ws.release(mark);
    return AID_OK;
}

// backward of everything that produces the modulations: adaLN Linear stack, SiLU(cond), obs encoder, time embeddings.
// q != null: the weight / bias gradients are recorded (every adjoint they read has a buffer of its own); the caller flushes, then releases ws.
int score_cond_backward(const AidScoreNet& n, const AidScoreNet& gr, int B, const ScoreActs& a, const float* dmod, const float* obs,
                        const float* drop_mask, const float* tvals, int t_mode, Workspace& ws, hipStream_t s, float* d_obs, GradQueue* q) {
    const int D = n.D, TE = n.TE, NA = n_ada(n), ldmod = NA * 2 * D;
    const size_t mark = ws.mark();
    float* dcond = ws.f32((size_t)B * D); float* e1 = ws.f32((size_t)B * D); float* e2 = ws.f32((size_t)B * D); float* e3 = ws.f32((size_t)B * D);
    float* e4 = ws.f32((size_t)B * D); float* e5 = ws.f32((size_t)B * D); float* f1 = ws.f32((size_t)B * D);
    float* dyeff = ws.f32((size_t)B * D); float* dte1 = ws.f32((size_t)B * 2 * D); float* dc = ws.f32((size_t)B * TE);
    float* dc_b = ws.f32((size_t)B * TE); float* dsin = ws.f32((size_t)B * TE);
    AID_WS_OK(ws);
    if (score_ada_contiguous(n) && score_ada_contiguous(gr)) {
        AID_TRY(lin_rev(n.blk[0].ada1_w, gr.blk[0].ada1_w, gr.blk[0].ada1_b, ldmod, D, B, a.sc, D, nullptr, dmod, nullptr, dcond, nullptr, true, s, q));
    } else {
        for (int i = 0; i < NA; ++i) {
            const float* w = i == NA - 1 ? n.adaf_w : ((i & 1) ? n.blk[i / 2].ada2_w : n.blk[i / 2].ada1_w);
            float* gw = i == NA - 1 ? gr.adaf_w : ((i & 1) ? gr.blk[i / 2].ada2_w : gr.blk[i / 2].ada1_w);
            float* gb = i == NA - 1 ? gr.adaf_b : ((i & 1) ? gr.blk[i / 2].ada2_b : gr.blk[i / 2].ada1_b);
            const float* dm = dmod + (size_t)i * 2 * D;
            if (q) AID_TRY(q->add(gw, D, 2 * D, D, dm, ldmod, a.sc, D, B, gb));
            else {
                AID_TRY(linear_bwd_weight(dm, ldmod, a.sc, D, gw, B, 2 * D, D, s, 1));
                AID_TRY(colsum(dm, ldmod, gb, B, 2 * D, 1, s));
            }
            AID_TRY(linear_bwd_data(dm, ldmod, w, D, dcond, D, B, 2 * D, s, i > 0));
        }
    }
    AID_TRY(act_bwd_dual(a.cond, dcond, nullptr, nullptr, ACT_SILU, (size_t)B * D, s));
    // ---- obs encoder (score_networks.py:49-59), reverse
    AID_TRY(ln_bwd_rows(dcond, D, a.o2, D, a.st_o2, n.oe8_g, n.oe8_b, ACT_NONE, nullptr, e1, D, nullptr, B, D, s));
    AID_TRY(ln_param_grad(dcond, a.o2, D, a.st_o2, gr.oe8_g, gr.oe8_b, B, D, s));
    AID_TRY(lin_rev(n.oe7_w, gr.oe7_w, gr.oe7_b, D, D, B, a.x2, D, nullptr, e1, nullptr, e2, nullptr, true, s, q));
    AID_TRY(ln_bwd_rows(e2, D, a.o1, D, a.st_o1, n.oe5_g, n.oe5_b, ACT_SILU, nullptr, e3, D, dyeff, B, D, s));
    AID_TRY(ln_param_grad(dyeff, a.o1, D, a.st_o1, gr.oe5_g, gr.oe5_b, B, D, s));
    AID_TRY(lin_rev(n.oe4_w, gr.oe4_w, gr.oe4_b, D, D, B, a.x1, D, nullptr, e3, nullptr, e4, nullptr, true, s, q));
    AID_TRY(ln_bwd_rows(e4, D, a.o0, D, a.st_o0, n.oe1_g, n.oe1_b, ACT_SILU, drop_mask, e5, D, dyeff, B, D, s));
    AID_TRY(ln_param_grad(dyeff, a.o0, D, a.st_o0, gr.oe1_g, gr.oe1_b, B, D, s));
    AID_TRY(lin_rev(n.oe0_w, gr.oe0_w, gr.oe0_b, D, n.O, B, obs, n.O, nullptr, e5, nullptr, nullptr, nullptr, true, s, q));
    if (d_obs) AID_TRY(linear_bwd_data(e5, D, n.oe0_w, n.O, d_obs, n.O, B, D, s, 1));   // pixel mode: the observation is the encoder output
    // ---- time embeddings (score_networks.py:119-141), dtemb = dcond
    if (t_mode != AID_T_DISCRETE) {
        AID_TRY(dot_accum(dcond, a.c2, (size_t)B * D, 1.f, gr.time_scale, s));
        AID_TRY(scale_by_ptr(dcond, n.time_scale, f1, (size_t)B * D, s));
        AID_TRY(lin_rev(n.cte4_w, gr.cte4_w, gr.cte4_b, D, TE, B, a.c1, TE, nullptr, f1, nullptr, dc, nullptr, true, s, q));
        AID_TRY(act_bwd_dual(a.c1_pre, dc, nullptr, nullptr, ACT_SILU, (size_t)B * TE, s));
        AID_TRY(lin_rev(n.cte2_w, gr.cte2_w, gr.cte2_b, TE, TE, B, a.c0, TE, nullptr, dc, nullptr, dc_b, nullptr, true, s, q));
        AID_TRY(act_bwd_dual(a.c0_pre, dc_b, nullptr, nullptr, ACT_SILU, (size_t)B * TE, s));
        AID_TRY(lin_rev(n.cte0_w, gr.cte0_w, gr.cte0_b, TE, 1, B, a.cin, 1, nullptr, dc_b, nullptr, nullptr, nullptr, true, s, q));
    }
    AID_TRY(lin_rev(n.te3_w, gr.te3_w, gr.te3_b, D, 2 * D, B, a.te1, 2 * D, nullptr, dcond, nullptr, dte1, nullptr, true, s, q));
    AID_TRY(act_bwd_dual(a.te1_pre, dte1, nullptr, nullptr, ACT_SILU, (size_t)B * 2 * D, s));
    AID_TRY(lin_rev(n.te1_w, gr.te1_w, gr.te1_b, 2 * D, TE, B, a.sin_e, TE, nullptr, dte1, nullptr, dsin, nullptr, true, s, q));
    AID_TRY(sinusoid_bwd(dsin, tvals, n.te_freq_scale, 999.0f, t_mode, gr.te_freq_scale, B, TE, s));
    if (!q) ws.release(mark);
    return AID_OK;
}

size_t score_cond_backward_ws_bytes(const AidScoreNet& n, int B) {
    return sizeof(float) * (size_t)B * (10 * (size_t)n.D + 3 * (size_t)n.TE) + 16 * 256;
}

}  // namespace aid
