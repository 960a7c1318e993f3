"""The chain's three products at 256 rows: gemm_mid (LDS-staged 32 x 64 tiles, split-K slabs) against gemm_w4 (32 x 16 tiles, waves split K, operands straight
to registers), each as a hipGraph of 240 dependent launches over 24 cold weight matrices (what a reverse loop looks like to the kernel)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
dev = "cuda"
def bench(fn, n=240):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(24): fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n): fn(i % 24)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / n * 1e6
for M in (256, 128, 32):
    for (N, K, act, ks_mid, ks_w4, name) in ((512, 512, "none", 4, 2, "x Wov^T"), (2048, 512, "gelu", 1, 1, "x W1^T + GELU"), (512, 2048, "none", 4, 4, "u W2^T")):
        A = torch.randn(M, K, device=dev); Ws = [torch.randn(N, K, device=dev) / K ** 0.5 for _ in range(24)]; b = torch.randn(N, device=dev)
        if M > 128 or True:
            try:
                t_mid = bench(lambda i: nat.gemm_mid(A, Ws[i], b if ks_mid == 1 else None, act if ks_mid == 1 else "none", ksplit=ks_mid))
            except Exception as e:
                t_mid = float("nan")
        t_w4 = bench(lambda i: nat.gemm_w4(A, Ws[i], b if ks_w4 == 1 else None, act if ks_w4 == 1 else "none", ksplit=ks_w4))
        print(f"M={M:4d} {name:14s} N={N} K={K}: gemm_mid (ksplit {ks_mid}) {t_mid:6.2f} us   gemm_w4 (ksplit {ks_w4}) {t_w4:6.2f} us", flush=True)
