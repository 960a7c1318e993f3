// Shared pieces of the full-row GEMM kernels (gemm_rowln.hip: v_mfma_f32_32x32x2_f32; gemm_rowln_x6.hip: the same product as six bf16 MFMA terms).
#pragma once
#include "aid_common.h"

namespace aid {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RowLnArgs {
    const float* A; int lda;             // [M][K] activations
    const float* W; int ldw;             // [BN][K] torch Linear weight
    const float* bias;                   // [BN] or null
    const float* gamma; const float* beta;   // LayerNorm affine, [BN] each (null => 1 / 0)
    const float* add; int ldadd;         // optional skip, added after the activation (and after the mask)
    const float* mask; int ldmask;       // optional multiplicative (dropout) mask applied to the activation
    float* Y; int ldy;
    int M, K;
    float eps;
};

template <int BKT> struct RowSw {
    static constexpr int CPR = BKT / 4;          // 16-B chunks per row
    static constexpr int RPB = 64 / BKT;         // rows per 256-B bank row
    __device__ static __forceinline__ int off(int row, int chunk) { return row * BKT + (((chunk ^ ((row / RPB) % CPR))) << 2); }
};

template <int ACT> __device__ __forceinline__ float rowln_act(float x) {
    if constexpr (ACT == ACT_RELU) return x > 0.f ? x : 0.f;
    else if constexpr (ACT == ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    else if constexpr (ACT == ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + expf(-x));
    else return x;
}

// 32 per-lane values (one per accumulator row) -> lane li (of each 32-lane half) ends with the half-wide sum of value li.
__device__ __forceinline__ float fold32(float (&x)[32], int li) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { const bool up = li & 16; const float recv = __shfl_xor(up ? x[i] : x[i + 16], 16, 64); x[i] = (up ? x[i + 16] : x[i]) + recv; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const bool up = li & 8; const float recv = __shfl_xor(up ? x[i] : x[i + 8], 8, 64); x[i] = (up ? x[i + 8] : x[i]) + recv; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const bool up = li & 4; const float recv = __shfl_xor(up ? x[i] : x[i + 4], 4, 64); x[i] = (up ? x[i + 4] : x[i]) + recv; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const bool up = li & 2; const float recv = __shfl_xor(up ? x[i] : x[i + 2], 2, 64); x[i] = (up ? x[i + 2] : x[i]) + recv; }
    { const bool up = li & 1; const float recv = __shfl_xor(up ? x[0] : x[1], 1, 64); x[0] = (up ? x[1] : x[0]) + recv; }
    return x[0];
}


// gemm_rowln_x6.hip: the same contract on v_mfma_f32_32x32x16_bf16 with both operands split three ways (see that file).  true when the shape is taken.
bool rowln_x6_enabled();
bool rowln_x6_ok(int N, int K, bool ln);
int launch_rowln_x6(const RowLnArgs& g, int N, int act, bool ln, hipStream_t s);

}  // namespace aid
