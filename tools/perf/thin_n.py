"""Thin-N / thin-K NT GEMMs of the EFE phase (N = 17, 1; K = 17) under each kernel the dispatcher could pick."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (M, N, K) in ((25600, 17, 512), (25600, 1, 512), (25600, 6, 256), (2560, 17, 512), (25600, 1024, 17), (25600, 128, 17)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.zeros(M, N, device="cuda"); b = torch.randn(N, device="cuda")
    row = []
    for code in (0, 1, 14, 15, 4, 9, 6):
        nat.gemm_force_kernel(code)
        try:
            dt = t(lambda: nat.gemm(A, W, C, M, N, K, K, K, N, bias=b, act="relu"))
            row.append(f"{code}:{dt*1e6:6.1f}us")
        except Exception as e:
            row.append(f"{code}: n/a")
        nat.gemm_force_kernel(0)
    print(f"M={M} N={N} K={K}: " + " ".join(row) + f"   bytes {4*(M*K+N*K+M*N)/1e6:.0f} MB")
