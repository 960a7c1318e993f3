// Counter-based device RNG (Philox4x32-10, Salmon et al. SC'11) for the production path: every
// stochastic op of the hot path (randn / rand / dropout masks; reference: torch.randn_like etc. at
// aid/core/diffusion.py:192,236, active_inference.py:449,577-580,1011) reads its noise from a tensor;
// when the caller passes NULL the host code fills that tensor here.  element i of a fill uses
// counter (offset + i/4), key = seed, so results do not depend on grid shape.
#include "aid_common.h"

namespace aid {

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// kind: 0 normal, 1 uniform [0,1), 2 dropout keep-mask scaled by 1/(1-p)
__global__ void philox_fill_kernel(float* __restrict__ out, size_t n, int kind, float p, uint64_t seed, uint64_t offset) {
    const size_t nq = (n + 3) / 4;
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; q < nq; q += stride) {
        const uint64_t ctr = offset + q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x41494421u, 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        float v[4];
        if (kind == 0) {
            const float r0 = sqrtf(-2.0f * logf(u01(c[0]))), r1 = sqrtf(-2.0f * logf(u01(c[2])));
            const float a0 = 6.283185307179586f * u01(c[1]), a1 = 6.283185307179586f * u01(c[3]);
            v[0] = r0 * cosf(a0); v[1] = r0 * sinf(a0); v[2] = r1 * cosf(a1); v[3] = r1 * sinf(a1);
        } else if (kind == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
        } else {
            const float sc = 1.0f / (1.0f - p);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)(c[j] >> 8) * (1.0f / 16777216.0f) >= p) ? sc : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q + j < n) out[4 * q + j] = v[j];
    }
}

int philox_fill(float* out, size_t n, int kind, float p, uint64_t seed, uint64_t offset, hipStream_t s) {
    if (n == 0) return AID_OK;
    AID_REQUIRE(out, AID_E_NULL, "philox_fill: null output");
    const size_t nq = (n + 3) / 4;
    const int blocks = (int)((nq + 255) / 256 > 4096 ? 4096 : (nq + 255) / 256);
    hipLaunchKernelGGL(philox_fill_kernel, dim3(blocks), dim3(256), 0, s, out, n, kind, p, seed, offset);
    AID_LAUNCH_CHECK();
    return AID_OK;
}

}  // namespace aid
