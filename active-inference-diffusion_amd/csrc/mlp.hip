#include "mlp.h"

namespace aid {

// ---------------------------------------------------------------------------------------------- small strided helpers
__global__ void copy_cols_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int M, int nc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * nc) return;
    const int r = i / nc, c = i % nc;
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];   // lds == 0 broadcasts one row
}
int copy_cols(const float* src, int lds, float* dst, int ldd, int M, int nc, hipStream_t s) {
    if (M * nc == 0) return AID_OK;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((M * nc + 255) / 256), dim3(256), 0, s, src, lds, dst, ldd, M, nc);
    AID_LAUNCH_CHECK(); return AID_OK;
}
// dst[r, c] = alpha * a[r*lda + c] + beta * b[r*ldb + c]
__global__ void axpby_cols_kernel(const float* __restrict__ a, int lda, float alpha, const float* __restrict__ b, int ldb, float beta,
                                  float* __restrict__ dst, int ldd, int M, int nc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * nc) return;
    const int r = i / nc, c = i % nc;
    float v = alpha * a[(size_t)r * lda + c];
    if (b) v += beta * b[(size_t)r * ldb + c];
    dst[(size_t)r * ldd + c] = v;
}
int axpby_cols(const float* a, int lda, float alpha, const float* b, int ldb, float beta, float* dst, int ldd, int M, int nc, hipStream_t s) {
    if (M * nc == 0) return AID_OK;
    hipLaunchKernelGGL(axpby_cols_kernel, dim3((M * nc + 255) / 256), dim3(256), 0, s, a, lda, alpha, b, ldb, beta, dst, ldd, M, nc);
    AID_LAUNCH_CHECK(); return AID_OK;
}

// ---------------------------------------------------------------------------------------------- LLA block
int lla_alloc(int M, int out, Workspace& ws, LlaSave& sv) {
    sv.lin = ws.f32((size_t)M * out); sv.stats = ws.f32((size_t)M * 2); sv.y = ws.f32((size_t)M * out);
    return AID_OK;
}
int lla_fwd(const AidLinear& l, const AidLN& ln, int act, const float* mask, const float* x, int ldx, int M, LlaSave& sv, hipStream_t s,
            const float* add, bool need_lin) {
    // K <= 32 (first layer of a net): one fused pass; the pre-LN values and statistics are written only when a backward pass will read them
    if (lin_smallk_ln_act_ok(l.in, l.out, act, l.w, sv.y, need_lin ? sv.lin : nullptr, mask, l.out) && ((((uintptr_t)add | (uintptr_t)l.b | (uintptr_t)ln.g | (uintptr_t)ln.b) & 15) == 0))
        return lin_smallk_ln_act(x, ldx, l.in, l.w, l.b, ln.g, ln.b, act, mask, add, need_lin ? sv.lin : nullptr, sv.y, l.out, need_lin ? sv.stats : nullptr, M, l.out, s);
    // inference on tall inputs (the estimator's 25 600+ sample rows): the LayerNorm lives in the GEMM's epilogue, nothing but y is written
    if (!need_lin && linear_ln_act_rows_ok(M, l.out, l.in, act, x, ldx, l.w, add, l.out, sv.y, l.out))
        return linear_ln_act_rows(x, ldx, l.w, l.b, ln.g, ln.b, act, add, l.out, sv.y, l.out, M, l.out, l.in, s, mask, l.out);
    AID_TRY(linear_fwd(x, ldx, l.w, l.b, sv.lin, l.out, M, l.out, l.in, ACT_NONE, s));
    AID_TRY(ln_act(sv.lin, l.out, ln.g, ln.b, act, mask, sv.y, l.out, sv.stats, M, l.out, s, add));
    return AID_OK;
}
int lla_bwd(const AidLinear& l, const AidLN& ln, const AidLinear* gl, const AidLN* gln, int act, const float* mask, const float* x,
            int ldx, int M, const LlaSave& sv, const float* dy, float* dx, int lddx, Workspace& ws, hipStream_t s, float* dlin_keep, float* dyeff_keep) {
    const size_t mark = ws.mark();
    float* dlin = dlin_keep ? dlin_keep : ws.f32((size_t)M * l.out);
    float* dyeff = dyeff_keep ? dyeff_keep : (gln ? ws.f32((size_t)M * l.out) : nullptr);
    AID_WS_OK(ws);
    AID_TRY(ln_bwd_rows(dy, l.out, sv.lin, l.out, sv.stats, ln.g, ln.b, act, mask, dlin, l.out, dyeff, M, l.out, s));
    if (gln && !dyeff_keep) AID_TRY(ln_param_grad(dyeff, sv.lin, l.out, sv.stats, gln->g, gln->b, M, l.out, s, &ws));
    if (gl && !dlin_keep) {
        AID_TRY(linear_bwd_weight(dlin, l.out, x, ldx, gl->w, M, l.out, l.in, s, 1));
        if (gl->b) AID_TRY(colsum(dlin, l.out, gl->b, M, l.out, 1, s, &ws));
    }
    if (dx) AID_TRY(linear_bwd_data(dlin, l.out, l.w, l.in, dx, lddx, M, l.out, s));
    ws.release(mark);
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------- LA block
int la_alloc(int M, int out, bool need_pre, Workspace& ws, LaSave& sv) {
    sv.pre = need_pre ? ws.f32((size_t)M * out) : nullptr; sv.y = ws.f32((size_t)M * out);
    return AID_OK;
}
int la_fwd(const AidLinear& l, int act, const float* mask, const float* x, int ldx, int M, LaSave& sv, int ldy, hipStream_t s) {
    return linear_fwd(x, ldx, l.w, l.b, sv.y, ldy, M, l.out, l.in, act, s, sv.pre, nullptr, 0, mask);
}
int la_bwd(const AidLinear& l, const AidLinear* gl, int act, const float* mask, const float* x, int ldx, int M, const LaSave& sv,
           float* dy, int lddy, float* dx, int lddx, hipStream_t s, int accumulate_dx, Workspace* ws, bool defer_params) {
    AID_REQUIRE(lddy == l.out, AID_E_SHAPE, "la_bwd: dy must be contiguous");
    if (mask) AID_TRY(mul_elem(dy, mask, dy, (size_t)M * l.out, s));
    if (act != ACT_NONE) AID_TRY(act_bwd_dual(sv.pre, dy, nullptr, nullptr, act, (size_t)M * l.out, s));
    if (gl && !defer_params) {
        AID_TRY(linear_bwd_weight(dy, l.out, x, ldx, gl->w, M, l.out, l.in, s, 1));
        if (gl->b) AID_TRY(colsum(dy, l.out, gl->b, M, l.out, 1, s, ws));
    }
    if (dx) AID_TRY(linear_bwd_data(dy, l.out, l.w, l.in, dx, lddx, M, l.out, s, accumulate_dx));
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------- decoder (forward only)
size_t decoder_ws_bytes(const AidDecoderNet& d, int M) {
    // tall inference inputs whose stages all take a fused Linear-LayerNorm path keep one buffer per stage (decoder_fwd); anything else two
    auto wide = [](int n) { return n == 512 || n == 1024; };
    const bool lean = M >= 4096 && wide(d.s[1].out) && wide(d.s[2].out) && d.s[1].in % 4 == 0 && d.s[2].in % 4 == 0 && d.s[1].in >= 64 && d.s[2].in >= 64 &&
                      (d.s[0].in <= 32 ? d.s[0].out % 256 == 0 : (wide(d.s[0].out) && d.s[0].in % 4 == 0 && d.s[0].in >= 64));
    return sizeof(float) * (size_t)M * ((size_t)(lean ? 1 : 2) * (d.s[0].out + d.s[1].out + d.s[2].out) + 8) + 64 * 256;
}
int decoder_fwd(const AidDecoderNet& d, int M, const float* z, int ldz, const float* const masks[3], float* out, int ldo, Workspace& ws,
                hipStream_t s) {
    const size_t mark = ws.mark();
    LlaSave s0, s1, s2;
    // inference on tall inputs writes nothing but each stage's output (LayerNorm in the GEMM epilogue): no pre-LN buffers then (at the
    // estimator's 640 000 rows they would be 6.5 GB)
    const bool tall = M >= 4096;
    if (!tall) { lla_alloc(M, d.s[0].out, ws, s0); lla_alloc(M, d.s[1].out, ws, s1); lla_alloc(M, d.s[2].out, ws, s2); }
    else {
        s0.y = ws.f32((size_t)M * d.s[0].out); s1.y = ws.f32((size_t)M * d.s[1].out); s2.y = ws.f32((size_t)M * d.s[2].out);
        // a stage gets pre-LN / statistics buffers only if it cannot take one of lla_fwd's fused paths (which write y alone)
        const float* m0 = masks ? masks[0] : nullptr;
        const bool f0 = (lin_smallk_ln_act_ok(d.s[0].in, d.s[0].out, ACT_SILU, d.s[0].w, s0.y, nullptr, m0, d.s[0].out) &&
                         ((((uintptr_t)d.s[0].b | (uintptr_t)d.ln[0].g | (uintptr_t)d.ln[0].b) & 15) == 0)) ||
                        linear_ln_act_rows_ok(M, d.s[0].out, d.s[0].in, ACT_SILU, z, ldz, d.s[0].w, nullptr, 0, s0.y, d.s[0].out);
        const bool f1 = linear_ln_act_rows_ok(M, d.s[1].out, d.s[1].in, ACT_SILU, s0.y, d.s[0].out, d.s[1].w, s0.y, d.s[0].out, s1.y, d.s[1].out);
        const bool f2 = linear_ln_act_rows_ok(M, d.s[2].out, d.s[2].in, ACT_SILU, s1.y, d.s[1].out, d.s[2].w, nullptr, 0, s2.y, d.s[2].out);
        s0.lin = f0 ? nullptr : ws.f32((size_t)M * d.s[0].out); s0.stats = f0 ? nullptr : ws.f32((size_t)M * 2);
        s1.lin = f1 ? nullptr : ws.f32((size_t)M * d.s[1].out); s1.stats = f1 ? nullptr : ws.f32((size_t)M * 2);
        s2.lin = f2 ? nullptr : ws.f32((size_t)M * d.s[2].out); s2.stats = f2 ? nullptr : ws.f32((size_t)M * 2);
    }
    AID_WS_OK(ws);
    AID_TRY(lla_fwd(d.s[0], d.ln[0], ACT_SILU, masks ? masks[0] : nullptr, z, ldz, M, s0, s, nullptr, false));   // inference: nothing saved
    AID_TRY(lla_fwd(d.s[1], d.ln[1], ACT_SILU, masks ? masks[1] : nullptr, s0.y, d.s[0].out, M, s1, s, s0.y, false));   // h2 = stage(h1) + h1 (skip, :240)
    AID_TRY(lla_fwd(d.s[2], d.ln[2], ACT_SILU, masks ? masks[2] : nullptr, s1.y, d.s[1].out, M, s2, s, nullptr, false));
    AID_TRY(linear_fwd(s2.y, d.s[2].out, d.out.w, d.out.b, out, ldo, M, d.out.out, d.out.in, ACT_NONE, s));
    ws.release(mark);
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------- reward predictor
int reward_fwd(const AidRewardNet& r, int M, const float* z, int ldz, RewardSave& sv, Workspace& ws, hipStream_t s) {
    lla_alloc(M, r.l0.out, ws, sv.s0); la_alloc(M, r.l3.out, true, ws, sv.s1); sv.out = ws.f32((size_t)M * 2);
    AID_WS_OK(ws);
    AID_TRY(lla_fwd(r.l0, r.ln, ACT_RELU, nullptr, z, ldz, M, sv.s0, s));
    AID_TRY(la_fwd(r.l3, ACT_RELU, nullptr, sv.s0.y, r.l0.out, M, sv.s1, r.l3.out, s));
    AID_TRY(linear_fwd(sv.s1.y, r.l3.out, r.l5.w, r.l5.b, sv.out, 2, M, 2, r.l5.in, ACT_NONE, s));
    return AID_OK;
}
int reward_bwd_data(const AidRewardNet& r, int M, const float* z, int ldz, const RewardSave& sv, float* d_out, float* dz, int lddz,
                    Workspace& ws, hipStream_t s) {
    const size_t mark = ws.mark();
    float* d1 = ws.f32((size_t)M * r.l3.out); float* d0 = ws.f32((size_t)M * r.l0.out);
    AID_WS_OK(ws);
    AID_TRY(linear_bwd_data(d_out, 2, r.l5.w, r.l5.in, d1, r.l3.out, M, 2, s));
    AID_TRY(la_bwd(r.l3, nullptr, ACT_RELU, nullptr, sv.s0.y, r.l0.out, M, sv.s1, d1, r.l3.out, d0, r.l0.out, s));
    AID_TRY(lla_bwd(r.l0, r.ln, nullptr, nullptr, ACT_RELU, nullptr, z, ldz, M, sv.s0, d0, dz, lddz, ws, s));
    ws.release(mark);
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------- dynamics
int dynamics_fwd(const AidDynamicsNet& d, int M, const float* z, int ldz, const float* a, int lda, float* next_mean, DynSave& sv,
                 Workspace& ws, hipStream_t s) {
    const int L = d.L, A = d.A, X = L + A;
    sv.x = ws.f32((size_t)M * X);
    for (int i = 0; i < 3; ++i) lla_alloc(M, d.net[i].out, ws, sv.s[i]);
    sv.out = ws.f32((size_t)M * L);
    AID_WS_OK(ws);
    AID_TRY(copy_cols(z, ldz, sv.x, X, M, L, s));
    AID_TRY(copy_cols(a, lda, sv.x + L, X, M, A, s));
    const float* x = sv.x; int ldx = X;
    for (int i = 0; i < 3; ++i) {
        AID_TRY(lla_fwd(d.net[i], d.ln[i], ACT_RELU, nullptr, x, ldx, M, sv.s[i], s));
        x = sv.s[i].y; ldx = d.net[i].out;
    }
    AID_TRY(linear_fwd(x, ldx, d.out.w, d.out.b, sv.out, L, M, L, d.out.in, ACT_NONE, s));
    // next_mean = z + (z + net) : LatentDynamicsModel residual + predict_next_latent residual (D7)
    AID_TRY(axpby_cols(z, ldz, 2.0f, sv.out, L, 1.0f, next_mean, L, M, L, s));
    return AID_OK;
}
int dynamics_bwd(const AidDynamicsNet& d, const AidDynamicsNet* g, int M, const DynSave& sv, const float* d_next, float* dz_out,
                 float* da_out, Workspace& ws, hipStream_t s, GradQueue* q) {
    const int L = d.L, A = d.A, X = L + A, D = d.net[0].out;
    // q: the four weight / bias gradients are recorded (their adjoints kept in buffers that outlive this call) and contracted by the caller's one grouped launch
    float* dl[3] = {nullptr, nullptr, nullptr};
    if (g && q) { for (int i = 0; i < 3; ++i) dl[i] = ws.f32((size_t)M * D); }
    const size_t mark = ws.mark();
    float* dA = ws.f32((size_t)M * D); float* dB = ws.f32((size_t)M * D); float* dx = ws.f32((size_t)M * X);
    AID_WS_OK(ws);
    if (g && q) AID_TRY(q->add(g->out.w, D, L, D, d_next, L, sv.s[2].y, D, M, g->out.b));
    else if (g) {
        AID_TRY(linear_bwd_weight(d_next, L, sv.s[2].y, D, g->out.w, M, L, D, s, 1));
        AID_TRY(colsum(d_next, L, g->out.b, M, L, 1, s));
    }
    AID_TRY(linear_bwd_data(d_next, L, d.out.w, D, dA, D, M, L, s));
    AID_TRY(lla_bwd(d.net[2], d.ln[2], g ? &g->net[2] : nullptr, g ? &g->ln[2] : nullptr, ACT_RELU, nullptr, sv.s[1].y, D, M, sv.s[2], dA, dB, D, ws, s, dl[2]));
    AID_TRY(lla_bwd(d.net[1], d.ln[1], g ? &g->net[1] : nullptr, g ? &g->ln[1] : nullptr, ACT_RELU, nullptr, sv.s[0].y, D, M, sv.s[1], dB, dA, D, ws, s, dl[1]));
    const bool need_dx = dz_out || da_out;
    AID_TRY(lla_bwd(d.net[0], d.ln[0], g ? &g->net[0] : nullptr, g ? &g->ln[0] : nullptr, ACT_RELU, nullptr, sv.x, X, M, sv.s[0], dA, need_dx ? dx : nullptr, X, ws, s, dl[0]));
    if (g && q) {
        AID_TRY(q->add(g->net[2].w, D, D, D, dl[2], D, sv.s[1].y, D, M, g->net[2].b));
        AID_TRY(q->add(g->net[1].w, D, D, D, dl[1], D, sv.s[0].y, D, M, g->net[1].b));
        AID_TRY(q->add(g->net[0].w, X, D, X, dl[0], D, sv.x, X, M, g->net[0].b));
    }
    if (dz_out) AID_TRY(axpby_cols(d_next, L, 2.0f, dx, X, 1.0f, dz_out, L, M, L, s));
    if (da_out) AID_TRY(copy_cols(dx + L, X, da_out, A, M, A, s));
    ws.release(mark);
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------- value
int value_fwd(const AidValueNet& v, int M, const float* state, int lds, const float* t, float* out, ValueSave& sv, Workspace& ws, hipStream_t s) {
    const int L = v.L, TE = v.TE, X = L + TE;
    sv.sin_e = ws.f32((size_t)M * TE); sv.te_pre = ws.f32((size_t)M * TE); sv.x = ws.f32((size_t)M * X);
    for (int i = 0; i < 3; ++i) lla_alloc(M, v.net[i].out, ws, sv.s[i]);
    sv.out = out;
    AID_WS_OK(ws);
    AID_TRY(sinusoid_embed(t, v.freq_scale, 1.0f, AID_T_DISCRETE, sv.sin_e, M, TE, s));
    AID_TRY(copy_cols(state, lds, sv.x, X, M, L, s));
    AID_TRY(linear_fwd(sv.sin_e, TE, v.te1.w, v.te1.b, sv.x + L, X, M, TE, TE, ACT_RELU, s));   // relu out > 0 <=> pre > 0 (backward)
    const float* x = sv.x; int ldx = X;
    for (int i = 0; i < 3; ++i) {
        AID_TRY(lla_fwd(v.net[i], v.ln[i], ACT_RELU, nullptr, x, ldx, M, sv.s[i], s));
        x = sv.s[i].y; ldx = v.net[i].out;
    }
    AID_TRY(linear_fwd(x, ldx, v.out.w, v.out.b, out, 1, M, 1, v.out.in, ACT_NONE, s));
    return AID_OK;
}
int value_bwd(const AidValueNet& v, const AidValueNet* g, int M, const float* t, const ValueSave& sv, const float* d_out, float* dstate,
              int ldds, Workspace& ws, hipStream_t s, GradQueue* q) {
    const int L = v.L, TE = v.TE, X = L + TE, D = v.net[0].out;
    // q: the five weight / bias gradients are recorded (their adjoints kept in buffers that outlive this call) and contracted by the caller's one grouped launch
    float* dl[3] = {nullptr, nullptr, nullptr}; float* dte_keep = nullptr;
    if (g && q) { for (int i = 0; i < 3; ++i) dl[i] = ws.f32((size_t)M * D); dte_keep = ws.f32((size_t)M * TE); }
    const size_t mark = ws.mark();
    float* dA = ws.f32((size_t)M * D); float* dB = ws.f32((size_t)M * D); float* dx = ws.f32((size_t)M * X);
    float* dte = dte_keep ? dte_keep : ws.f32((size_t)M * TE); float* dsin = ws.f32((size_t)M * TE);
    AID_WS_OK(ws);
    if (g && q) AID_TRY(q->add(g->out.w, D, 1, D, d_out, 1, sv.s[2].y, D, M, g->out.b));
    else if (g) {
        AID_TRY(linear_bwd_weight(d_out, 1, sv.s[2].y, D, g->out.w, M, 1, D, s, 1));
        AID_TRY(colsum(d_out, 1, g->out.b, M, 1, 1, s));
    }
    AID_TRY(linear_bwd_data(d_out, 1, v.out.w, D, dA, D, M, 1, s));
    AID_TRY(lla_bwd(v.net[2], v.ln[2], g ? &g->net[2] : nullptr, g ? &g->ln[2] : nullptr, ACT_RELU, nullptr, sv.s[1].y, D, M, sv.s[2], dA, dB, D, ws, s, dl[2]));
    AID_TRY(lla_bwd(v.net[1], v.ln[1], g ? &g->net[1] : nullptr, g ? &g->ln[1] : nullptr, ACT_RELU, nullptr, sv.s[0].y, D, M, sv.s[1], dB, dA, D, ws, s, dl[1]));
    const bool need_dx = dstate || g;
    AID_TRY(lla_bwd(v.net[0], v.ln[0], g ? &g->net[0] : nullptr, g ? &g->ln[0] : nullptr, ACT_RELU, nullptr, sv.x, X, M, sv.s[0], dA, need_dx ? dx : nullptr, X, ws, s, dl[0]));
    if (g && q) {
        AID_TRY(q->add(g->net[2].w, D, D, D, dl[2], D, sv.s[1].y, D, M, g->net[2].b));
        AID_TRY(q->add(g->net[1].w, D, D, D, dl[1], D, sv.s[0].y, D, M, g->net[1].b));
        AID_TRY(q->add(g->net[0].w, X, D, X, dl[0], D, sv.x, X, M, g->net[0].b));
    }
    if (dstate) AID_TRY(copy_cols(dx, X, dstate, ldds, M, L, s));
    if (g) {  // time-embedding branch: te = relu(te1(sinus(t)))
        AID_TRY(copy_cols(dx + L, X, dte, TE, M, TE, s));
        // relu'(pre) with pre stored at stride X inside sv.te_pre's buffer? (see value_fwd) -> use the output: relu out > 0 <=> pre > 0
        AID_TRY(copy_cols(sv.x + L, X, dsin, TE, M, TE, s));                       // dsin temporarily = relu output
        AID_TRY(act_bwd_dual(dsin, dte, nullptr, nullptr, ACT_RELU, (size_t)M * TE, s));
        if (q) AID_TRY(q->add(g->te1.w, TE, TE, TE, dte, TE, sv.sin_e, TE, M, g->te1.b));
        else {
            AID_TRY(linear_bwd_weight(dte, TE, sv.sin_e, TE, g->te1.w, M, TE, TE, s, 1));
            AID_TRY(colsum(dte, TE, g->te1.b, M, TE, 1, s));
        }
        AID_TRY(linear_bwd_data(dte, TE, v.te1.w, TE, dsin, TE, M, TE, s));
        AID_TRY(sinusoid_bwd(dsin, t, v.freq_scale, 1.0f, AID_T_DISCRETE, g->freq_scale, M, TE, s));
    }
    ws.release(mark);
    return AID_OK;
}

// ---------------------------------------------------------------------------------------------- policy
__global__ void policy_sample_kernel(const float* __restrict__ mean, const float* __restrict__ lstd, const float* __restrict__ eps,
                                     float* __restrict__ action, float* __restrict__ ent, float* __restrict__ logp, int M, int A) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    float e = 0.f, lp = 0.f;
    for (int a = 0; a < A; ++a) {
        const float ls = fminf(fmaxf(lstd[(size_t)r * A + a], -20.f), 2.f);
        const float sd = expf(ls), m = mean[(size_t)r * A + a];
        const float ep = eps ? eps[(size_t)r * A + a] : 0.f;
        const float act = m + sd * ep;
        action[(size_t)r * A + a] = act;
        e += 0.5f + 0.5f * 1.8378770664093453f + ls;                      // 0.5 + 0.5 log(2 pi) + log std
        const float d = act - m;
        lp += -(d * d) / (2.f * sd * sd) - ls - 0.9189385332046727f;      // log sqrt(2 pi)
    }
    ent[r] = e; logp[r] = lp;
}
__global__ void policy_sample_bwd_kernel(const float* __restrict__ lstd, const float* __restrict__ eps, const float* __restrict__ d_action,
                                         const float* __restrict__ ent_coef, float* __restrict__ dmean, float* __restrict__ dlstd, int M, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * A) return;
    const int r = i / A;
    const float raw = lstd[i];
    const float ls = fminf(fmaxf(raw, -20.f), 2.f);
    const float da = d_action ? d_action[i] : 0.f;
    dmean[i] = da;
    float dls = da * expf(ls) * (eps ? eps[i] : 0.f) + (ent_coef ? ent_coef[r] : 0.f);
    dlstd[i] = (raw >= -20.f && raw <= 2.f) ? dls : 0.f;
}

int policy_fwd(const AidPolicyNet& p, int M, const float* z, int ldz, const float* eps, PolicySave& sv, Workspace& ws, hipStream_t s) {
    const int D = p.D, A = p.A, H = p.mean0.out;
    if (!sv.prealloc) {
        lla_alloc(M, D, ws, sv.e0); sv.h = ws.f32((size_t)M * D);
        for (int i = 0; i < 3; ++i) lla_alloc(M, D, ws, sv.tr[i]);
        sv.h2 = ws.f32((size_t)M * D);
        la_alloc(M, H, true, ws, sv.m0); la_alloc(M, H, true, ws, sv.s0);
        sv.mean = ws.f32((size_t)M * A); sv.lstd = ws.f32((size_t)M * A); sv.action = ws.f32((size_t)M * A);
        sv.ent = ws.f32(M); sv.logp = ws.f32(M);
    }
    AID_WS_OK(ws);
    AID_TRY(lla_fwd(p.enc0, p.enc_ln, ACT_RELU, nullptr, z, ldz, M, sv.e0, s));
    AID_TRY(linear_fwd(sv.e0.y, D, p.enc3.w, p.enc3.b, sv.h, D, M, D, D, ACT_NONE, s));
    const float* x = sv.h;
    for (int i = 0; i < 3; ++i) { AID_TRY(lla_fwd(p.trunk[i], p.trunk_ln[i], ACT_RELU, nullptr, x, D, M, sv.tr[i], s)); x = sv.tr[i].y; }
    AID_TRY(axpby_cols(sv.h, D, 1.0f, sv.tr[2].y, D, 1.0f, sv.h2, D, M, D, s));   // h + trunk(h)
    AID_TRY(la_fwd(p.mean0, ACT_RELU, nullptr, sv.h2, D, M, sv.m0, H, s));
    AID_TRY(linear_fwd(sv.m0.y, H, p.mean2.w, p.mean2.b, sv.mean, A, M, A, H, ACT_NONE, s));
    AID_TRY(la_fwd(p.lstd0, ACT_RELU, nullptr, sv.h2, D, M, sv.s0, H, s));
    AID_TRY(linear_fwd(sv.s0.y, H, p.lstd2.w, p.lstd2.b, sv.lstd, A, M, A, H, ACT_NONE, s));
    hipLaunchKernelGGL(policy_sample_kernel, dim3((M + 255) / 256), dim3(256), 0, s, sv.mean, sv.lstd, eps, sv.action, sv.ent, sv.logp, M, A);
    AID_LAUNCH_CHECK();
    return AID_OK;
}

int policy_steps_alloc(const AidPolicyNet& p, int M, int H, Workspace& ws, PolicySteps& ps) {
    AID_REQUIRE(H >= 1 && H <= 16, AID_E_SHAPE, "policy_steps_alloc: H=%d", H);
    const int D = p.D, A = p.A, Hh = p.mean0.out;
    ps.H = H; ps.M = M;
    const size_t HM = (size_t)H * M;
    auto slices = [&](size_t width, float* PolicySave::* field) {
        float* base = ws.f32(HM * width);
        for (int t = 0; t < H; ++t) ps.sv[t].*field = base ? base + (size_t)t * M * width : nullptr;
    };
    auto lla_slices = [&](size_t width, LlaSave PolicySave::* blk) {
        float* lin = ws.f32(HM * width); float* st = ws.f32(HM * 2); float* y = ws.f32(HM * width);
        for (int t = 0; t < H; ++t) {
            LlaSave& b = ps.sv[t].*blk;
            b.lin = lin ? lin + (size_t)t * M * width : nullptr; b.stats = st ? st + (size_t)t * M * 2 : nullptr; b.y = y ? y + (size_t)t * M * width : nullptr;
        }
    };
    auto la_slices = [&](size_t width, LaSave PolicySave::* blk) {
        float* pre = ws.f32(HM * width); float* y = ws.f32(HM * width);
        for (int t = 0; t < H; ++t) { LaSave& b = ps.sv[t].*blk; b.pre = pre ? pre + (size_t)t * M * width : nullptr; b.y = y ? y + (size_t)t * M * width : nullptr; }
    };
    lla_slices(D, &PolicySave::e0); slices(D, &PolicySave::h);
    {   // tr[3] is an array member: slice it by hand
        for (int i = 0; i < 3; ++i) {
            float* lin = ws.f32(HM * D); float* st = ws.f32(HM * 2); float* y = ws.f32(HM * D);
            for (int t = 0; t < H; ++t) {
                LlaSave& b = ps.sv[t].tr[i];
                b.lin = lin ? lin + (size_t)t * M * D : nullptr; b.stats = st ? st + (size_t)t * M * 2 : nullptr; b.y = y ? y + (size_t)t * M * D : nullptr;
            }
        }
    }
    slices(D, &PolicySave::h2);
    la_slices(Hh, &PolicySave::m0); la_slices(Hh, &PolicySave::s0);
    slices(A, &PolicySave::mean); slices(A, &PolicySave::lstd); slices(A, &PolicySave::action); slices(1, &PolicySave::ent); slices(1, &PolicySave::logp);
    for (int t = 0; t < H; ++t) ps.sv[t].prealloc = true;
    ps.dmean = ws.f32(HM * A); ps.dlstd = ws.f32(HM * A); ps.dm0 = ws.f32(HM * Hh); ps.ds0 = ws.f32(HM * Hh);
    for (int i = 0; i < 3; ++i) ps.dtr[i] = ws.f32(HM * D);
    ps.de0 = ws.f32(HM * D); ps.dh = ws.f32(HM * D);
    for (int i = 0; i < 4; ++i) ps.dyeff[i] = ws.f32(HM * D);
    AID_WS_OK(ws);
    return AID_OK;
}

int policy_bptt_grads(const AidPolicyNet& p, const AidPolicyNet& g, const PolicySteps& ps, const float* z_all, int ldz, GradQueue& q, Workspace& ws) {
    const int D = p.D, A = p.A, Hh = p.mean0.out, R = ps.H * ps.M;
    const PolicySave& s0 = ps.sv[0];                      // slice 0 = base of every [H][M][.] buffer
    // LayerNorm gamma / beta: column sums of dyeff x_hat and dyeff over the H*M rows of all steps (3 launches per layer instead of 3 H)
    AID_TRY(ln_param_grad(ps.dyeff[0], s0.e0.lin, D, s0.e0.stats, g.enc_ln.g, g.enc_ln.b, R, D, q.s, &ws));
    for (int i = 0; i < 3; ++i) AID_TRY(ln_param_grad(ps.dyeff[1 + i], s0.tr[i].lin, D, s0.tr[i].stats, g.trunk_ln[i].g, g.trunk_ln[i].b, R, D, q.s, &ws));
    AID_TRY(q.add(g.mean2.w, Hh, A, Hh, ps.dmean, A, s0.m0.y, Hh, R, g.mean2.b));
    AID_TRY(q.add(g.lstd2.w, Hh, A, Hh, ps.dlstd, A, s0.s0.y, Hh, R, g.lstd2.b));
    AID_TRY(q.add(g.mean0.w, D, Hh, D, ps.dm0, Hh, s0.h2, D, R, g.mean0.b));
    AID_TRY(q.add(g.lstd0.w, D, Hh, D, ps.ds0, Hh, s0.h2, D, R, g.lstd0.b));
    AID_TRY(q.add(g.trunk[2].w, D, D, D, ps.dtr[2], D, s0.tr[1].y, D, R, g.trunk[2].b));
    AID_TRY(q.add(g.trunk[1].w, D, D, D, ps.dtr[1], D, s0.tr[0].y, D, R, g.trunk[1].b));
    AID_TRY(q.add(g.trunk[0].w, D, D, D, ps.dtr[0], D, s0.h, D, R, g.trunk[0].b));
    AID_TRY(q.add(g.enc3.w, D, D, D, ps.dh, D, s0.e0.y, D, R, g.enc3.b));
    AID_TRY(q.add(g.enc0.w, p.enc0.in, D, p.enc0.in, ps.de0, D, z_all, ldz, R, g.enc0.b));
    return AID_OK;
}

int policy_bwd(const AidPolicyNet& p, const AidPolicyNet* g, int M, const float* z, int ldz, const float* eps, const PolicySave& sv,
               const float* d_action, const float* ent_coef, float* dz, int lddz, Workspace& ws, hipStream_t s, const PolicySteps* steps, int t) {
    const int D = p.D, A = p.A, H = p.mean0.out;
    const size_t mark = ws.mark();
    const bool defer = steps != nullptr && g != nullptr;
    auto slot = [&](float* base, size_t width) { return base + (size_t)t * M * width; };
    float* dmean = defer ? slot(steps->dmean, A) : ws.f32((size_t)M * A); float* dlstd = defer ? slot(steps->dlstd, A) : ws.f32((size_t)M * A);
    float* dm0 = defer ? slot(steps->dm0, H) : ws.f32((size_t)M * H); float* ds0 = defer ? slot(steps->ds0, H) : ws.f32((size_t)M * H);
    float* dh2 = ws.f32((size_t)M * D); float* dA = ws.f32((size_t)M * D); float* dB = ws.f32((size_t)M * D);
    float* dhk = defer ? slot(steps->dh, D) : nullptr;
    AID_WS_OK(ws);
    hipLaunchKernelGGL(policy_sample_bwd_kernel, dim3((M * A + 255) / 256), dim3(256), 0, s, sv.lstd, eps, d_action, ent_coef, dmean, dlstd, M, A);
    AID_LAUNCH_CHECK();
    // heads
    if (g && !defer) {
        AID_TRY(linear_bwd_weight(dmean, A, sv.m0.y, H, g->mean2.w, M, A, H, s, 1)); AID_TRY(colsum(dmean, A, g->mean2.b, M, A, 1, s, &ws));
        AID_TRY(linear_bwd_weight(dlstd, A, sv.s0.y, H, g->lstd2.w, M, A, H, s, 1)); AID_TRY(colsum(dlstd, A, g->lstd2.b, M, A, 1, s, &ws));
    }
    AID_TRY(linear_bwd_data(dmean, A, p.mean2.w, H, dm0, H, M, A, s));
    AID_TRY(linear_bwd_data(dlstd, A, p.lstd2.w, H, ds0, H, M, A, s));
    AID_TRY(la_bwd(p.mean0, g ? &g->mean0 : nullptr, ACT_RELU, nullptr, sv.h2, D, M, sv.m0, dm0, H, dh2, D, s, 0, &ws, defer));
    AID_TRY(la_bwd(p.lstd0, g ? &g->lstd0 : nullptr, ACT_RELU, nullptr, sv.h2, D, M, sv.s0, ds0, H, dh2, D, s, 1, &ws, defer));
    // h2 = h + trunk(h)
    AID_TRY(lla_bwd(p.trunk[2], p.trunk_ln[2], g ? &g->trunk[2] : nullptr, g ? &g->trunk_ln[2] : nullptr, ACT_RELU, nullptr, sv.tr[1].y, D, M, sv.tr[2], dh2, dA, D, ws, s,
                    defer ? slot(steps->dtr[2], D) : nullptr, defer ? slot(steps->dyeff[3], D) : nullptr));
    AID_TRY(lla_bwd(p.trunk[1], p.trunk_ln[1], g ? &g->trunk[1] : nullptr, g ? &g->trunk_ln[1] : nullptr, ACT_RELU, nullptr, sv.tr[0].y, D, M, sv.tr[1], dA, dB, D, ws, s,
                    defer ? slot(steps->dtr[1], D) : nullptr, defer ? slot(steps->dyeff[2], D) : nullptr));
    float* dhv = defer ? dhk : dA;                       // dh = d(trunk input) + dh2 (the residual): kept per step when the gradients are deferred
    AID_TRY(lla_bwd(p.trunk[0], p.trunk_ln[0], g ? &g->trunk[0] : nullptr, g ? &g->trunk_ln[0] : nullptr, ACT_RELU, nullptr, sv.h, D, M, sv.tr[0], dB, dhv, D, ws, s,
                    defer ? slot(steps->dtr[0], D) : nullptr, defer ? slot(steps->dyeff[1], D) : nullptr));
    AID_TRY(add_inplace(dhv, dh2, (size_t)M * D, s));   // dh
    if (g && !defer) {
        AID_TRY(linear_bwd_weight(dA, D, sv.e0.y, D, g->enc3.w, M, D, D, s, 1)); AID_TRY(colsum(dA, D, g->enc3.b, M, D, 1, s, &ws));
    }
    AID_TRY(linear_bwd_data(dhv, D, p.enc3.w, D, dB, D, M, D, s));
    AID_TRY(lla_bwd(p.enc0, p.enc_ln, g ? &g->enc0 : nullptr, g ? &g->enc_ln : nullptr, ACT_RELU, nullptr, z, ldz, M, sv.e0, dB, dz, lddz, ws, s,
                    defer ? slot(steps->de0, D) : nullptr, defer ? slot(steps->dyeff[0], D) : nullptr));
    ws.release(mark);
    return AID_OK;
}

}  // namespace aid
