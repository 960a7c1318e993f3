// Accuracy probe: C = A B^T (A [M][K], B [N][K], fp32 data) on one wave per 32 x 32 tile, four ways:
//   0  v_mfma_f32_32x32x2_f32 chain (what the library runs)
//   1  three-way bf16 split of both operands, 6 products (hh, hm, mh, hl, lh, mm) on v_mfma_f32_32x32x16_bf16, ONE accumulator
//   2  the same with the leading product (hh) and the five corrections in separate accumulators, added at the end
//   3  plain bf16 (hh only) for scale
// against a float64 host product.  Prints max and RMS error relative to sum_k |a||b| (the natural scale of a dot product's rounding error).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}
template <int MODE>
__global__ void probe(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    f32x16 acc = {0}, acc2 = {0};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) {
            const float a = A[(size_t)(i0 + lane % 32) * K + k + lane / 32], b = B[(size_t)(j0 + lane % 32) * K + k + lane / 32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                split3(A[(size_t)(i0 + lane % 32) * K + k + 8 * (lane / 32) + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
                split3(B[(size_t)(j0 + lane % 32) * K + k + 8 * (lane / 32) + e], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
            }
            if (MODE == 3) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0); continue; }
            f32x16& small = MODE == 2 ? acc2 : acc;
            // smallest first within a k-step
            small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, small, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        if (MODE == 2) for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r / 4) * 8 + (lane / 32) * 4 + r % 4, col = lane % 32;
        C[(size_t)(i0 + row) * N + j0 + col] = acc[r];
    }
}
int main(int argc, char** argv) {
    const int M = 128, N = 128;
    for (int K : {512, 1024, 2048, 16384}) for (int dist = 0; dist < 2; ++dist) {
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        srand(1234 + K);
        auto rnd = [&]() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
        for (auto& x : A) x = (float)(dist ? fabs(rnd()) : rnd());       // dist 1: all-positive operands (no cancellation: a biased accumulate shows)
        for (auto& x : B) x = (float)(dist ? fabs(rnd()) : rnd());
        std::vector<double> ref((size_t)M * N), scale((size_t)M * N);
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { double s = 0, q = 0; for (int k = 0; k < K; ++k) { double p = (double)A[(size_t)i * K + k] * B[(size_t)j * K + k]; s += p; q += fabs(p); } ref[i * N + j] = s; scale[i * N + j] = q; }
        float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> C((size_t)M * N);
        for (int mode = 0; mode < 4; ++mode) {
            dim3 g(N / 32, M / 32);
            if (mode == 0) probe<0><<<g, 64>>>(dA, dB, dC, M, N, K); else if (mode == 1) probe<1><<<g, 64>>>(dA, dB, dC, M, N, K);
            else if (mode == 2) probe<2><<<g, 64>>>(dA, dB, dC, M, N, K); else probe<3><<<g, 64>>>(dA, dB, dC, M, N, K);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, sq = 0, bias = 0;
            for (size_t i = 0; i < C.size(); ++i) { const double e = ((double)C[i] - ref[i]) / scale[i]; mx = fmax(mx, fabs(e)); sq += e * e; bias += e; }
            printf("K=%5d %s mode %d (%s): max %.3e  rms %.3e  mean %.3e   (x sum|a||b|)\n", K, dist ? "positive" : "gaussian", mode,
                   mode == 0 ? "f32 mfma    " : mode == 1 ? "bf16x6 1 acc" : mode == 2 ? "bf16x6 2 acc" : "bf16 plain  ", mx, sqrt(sq / C.size()), bias / C.size());
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
