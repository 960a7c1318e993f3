#pragma once
#include "score_net.h"

namespace aid {

struct ScoreTanBlock { float *nx1d, *vd, *hmd, *nx2d, *a1d, *ud; };
// tangent activations (direction z_dot) of one score forward
struct ScoreTan {
    float* hd[AID_MAX_BLOCKS + 1];
    ScoreTanBlock blk[AID_MAX_BLOCKS];
    float *nxfd, *p0pd, *p0d, *rawd;
};
size_t score_tan_bytes(const AidScoreNet& n, int B);
int score_tan_alloc(const AidScoreNet& n, int B, Workspace& ws, ScoreTan& t);
int score_tangent_forward(const AidScoreNet& n, int B, const float* z_dot, const ScoreActs& a, ScoreTan& t, hipStream_t s);

// Reverse pass over the trunk (latent_proj .. output_proj).
//   t == null : ordinary backward with cotangent d_score on the score (null = ones); param grads into *gr when gr != null
//   t != null : adjoint of coef * sum(tangent score) (gradient-penalty pass); d_score ignored
// dz (optional) receives d/dz_t; dmod (optional) [B, n_ada*2D] receives the modulation gradients.
// q (optional): record the weight / bias gradients instead of launching them (gemm_group.hip); the caller flushes q, THEN releases the workspace mark
// it took before the call (the recorded problems read adjoints that live in `ws`).
size_t score_reverse_ws_bytes(const AidScoreNet& n, int B);
int score_reverse(const AidScoreNet& n, const AidScoreNet* gr, int B, const ScoreActs& a, const ScoreTan* t, const float* z,
                  const float* z_dot, const float* d_score, const float* tvals, int t_mode, float coef, float* dz, float* dmod,
                  Workspace& ws, hipStream_t s, GradQueue* q = nullptr);
size_t score_cond_backward_ws_bytes(const AidScoreNet& n, int B);
int score_cond_backward(const AidScoreNet& n, const AidScoreNet& gr, int B, const ScoreActs& a, const float* dmod, const float* obs,
                        const float* drop_mask, const float* tvals, int t_mode, Workspace& ws, hipStream_t s, float* d_obs = nullptr, GradQueue* q = nullptr);

}  // namespace aid
