// LatentScoreNetwork on gfx950: forward (with or without saved activations), reverse-diffusion loop,
// and (score_bwd.hip) backward / tangent / dual-backward passes.  Reference:
// aid/models/score_networks.py:12-291, aid/core/diffusion.py:176-255.
#pragma once
#include "aid_common.h"
#include "rowwise.h"

namespace aid {

int philox_fill(float* out, size_t n, int kind, float p, uint64_t seed, uint64_t offset, hipStream_t s);

struct ScoreBlockActs {
    float *h_in;    // [B,D] input of the block
    float *st1;     // [B,2] LN stats of h_in
    float *nx1;     // [B,D] adaLN1 output
    float *v;       // [B,D] V projection
    float *h_mid;   // [B,D] after attention residual
    float *st2;     // [B,2]
    float *nx2;     // [B,D]
    float *a1;      // [B,4D] pre-GELU
    float *u;       // [B,4D] GELU(a1)
};

// Every intermediate of one score forward.  With keep=false the per-block buffers alias.
struct ScoreActs {
    int B;
    // time path (rows = Bt: B, or 1 when t is shared)
    float *sin_e, *te1_pre, *te1, *t_sin, *cin, *c0_pre, *c0, *c1_pre, *c1, *c2, *temb;
    // obs path
    float *o0, *st_o0, *x1n, *x1, *o1, *st_o1, *x2, *o2, *st_o2;
    float *cond, *sc, *mod;
    float *h0;
    ScoreBlockActs blk[AID_MAX_BLOCKS];
    float *h_last, *st_f, *nxf, *p0_pre, *p0, *raw;
};

size_t score_acts_bytes(const AidScoreNet& n, int B, bool keep);
int score_acts_alloc(const AidScoreNet& n, int B, bool keep, Workspace& ws, ScoreActs& a);

// whether the 13 adaLN modulation weights/biases are contiguous (one fused GEMM) in this table
bool score_ada_contiguous(const AidScoreNet& n);
int score_validate(const AidScoreNet* n);

// temb[M,D] for times t[M]; mode AID_T_* or 2 (per-row by value).  Fills a.sin_e.. a.temb (rows M).
int score_time_embed(const AidScoreNet& n, int M, const float* t, int mode, ScoreActs& a, hipStream_t s);
// obs path up to the pre-mask activation x1n = SiLU(LN(Linear(obs)))  (mask-independent part)
int score_obs_stage0(const AidScoreNet& n, int B, const float* obs, ScoreActs& a, hipStream_t s);
// rest of the obs path + cond/sc + all 13 modulations.  temb row stride ld_temb (0 = shared row).
int score_cond(const AidScoreNet& n, int B, const float* drop_mask, const float* temb, int ld_temb, ScoreActs& a, hipStream_t s);
// out_proj(V_proj(x)) collapsed into one weight per block (valid while the weights do not change: inference loops)
// slab: [4][rows][D] scratch for the split-K partial products of the mid-M pipeline (null: not used)
// wlp_t: latent_proj weight transposed to [L][D] (folded into block 0's adaLN); p0_slabs: set by the trunk when it left the head's first
// layer as 4 raw slabs in `slab` ([4][rows][D/2]) for head2_slab_p_sample instead of writing p0
struct ScoreFused { float* wov[AID_MAX_BLOCKS]; float* bov[AID_MAX_BLOCKS]; float* slab = nullptr; float* slab2 = nullptr; float* wlp_t = nullptr; bool p0_slabs = false;
                    bool lp_done = false; };   // lp_done: block 0's adaLN (+ latent projection) of this step was produced by the previous step's reverse-step kernel (NextLp)
int score_fuse_attention(const AidScoreNet& n, Workspace& ws, ScoreFused& f, hipStream_t s);
// latent_proj, DiT blocks, final adaLN, output_proj -> a.raw [B,L]
int score_trunk(const AidScoreNet& n, int B, const float* z, ScoreActs& a, bool keep, hipStream_t s, ScoreFused* fused = nullptr, bool skip_last = false);

}  // namespace aid
