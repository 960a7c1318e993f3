// What a dependent kernel costs on this GPU whatever it does: chains of 2000 kernels on one stream, wall time per kernel, for
//   empty kernels of 1 / 64 / 256 / 1024 workgroups, a kernel that writes 0.5 MB (the chains' adaln_sum size), one that reads + writes it,
//   and the same chains replayed from a hipGraph.  Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty() {}
__global__ void k_write(float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = 1.f; }
__global__ void k_rw(const float* x, float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i] + 1.f; }
__global__ void __launch_bounds__(256) k_lds(float* y) { __shared__ float s[12288]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (s[255 - threadIdx.x] < 0.f) y[0] = 1.f; }
template <class F> static double chain(hipStream_t s, int n, F launch, bool graph) {
    hipGraphExec_t ge = nullptr;
    if (graph) {
        hipGraph_t g; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch(i);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    } else { for (int i = 0; i < 50; ++i) launch(i); hipStreamSynchronize(s); }
    auto t0 = std::chrono::steady_clock::now();
    if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n; ++i) launch(i);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    const int n = 131072; float *a, *b; hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMemset(a, 0, 4 * n);
    for (int graph = 0; graph < 2; ++graph) {
        for (int wg : {1, 64, 256, 1024}) printf("%s empty kernel, %4d workgroups x 256: %.2f us per kernel\n", graph ? "graph " : "stream", wg, chain(s, 2000, [&](int) { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, s); }, graph));
        printf("%s 48 KB LDS, 256 workgroups:          %.2f us per kernel\n", graph ? "graph " : "stream", chain(s, 2000, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, s, b); }, graph));
        printf("%s write 0.5 MB, 512 workgroups:        %.2f us per kernel\n", graph ? "graph " : "stream", chain(s, 2000, [&](int) { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, s, b, n); }, graph));
        printf("%s ping-pong read + write 0.5 MB:       %.2f us per kernel\n", graph ? "graph " : "stream", chain(s, 2000, [&](int i) { hipLaunchKernelGGL(k_rw, dim3(n / 256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n); }, graph));
    }
    return 0;
}
