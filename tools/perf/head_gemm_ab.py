"""The encoder head's data gradient (dX = dY W, W as stored) and weight gradient (dW = dY^T X) at 256 x 451 584 x 256: 64 x 64 against 128 x 128 tiles of gemm_big_kernel
(aid_gemm_force_kernel 9 / 12)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
B, L2, F = 256, 256, 451584
dy = torch.randn(B, L2, device="cuda"); W = torch.randn(L2, F, device="cuda") * 0.01; x = torch.randn(B, F, device="cuda")
dx = torch.empty(B, F, device="cuda"); dW = torch.zeros(L2, F, device="cuda")
fl = 2.0 * B * L2 * F
def t(fn, reps=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for code, nm in ((9, "64 x 64 tiles"), (12, "128 x 128 tiles"), (0, "automatic")):
    nat.gemm_force_kernel(code)
    a = t(lambda: nat.gemm(dy, W, dx, B, F, L2, L2, F, F, 0, 1))                      # dX[B,F] = dY[B,L2] W[L2,F]   (b_kmajor)
    b = t(lambda: nat.gemm(dy, x, dW, L2, F, B, L2, F, F, 1, 1, accumulate=1))        # dW[L2,F] += dY^T[L2,B] X[B,F]  (a_kmajor, b_kmajor)
    print(f"{nm:18s}: dX {a*1e3:6.3f} ms {fl/a/1e12:6.1f} TFLOP/s | dW {b*1e3:6.3f} ms {fl/b/1e12:6.1f} TFLOP/s", flush=True)
nat.gemm_force_kernel(0)
