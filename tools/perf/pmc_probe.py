import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
code = int(sys.argv[1])
M, N, K = 512, 512, 512
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
nat.gemm_force_kernel(code)
for _ in range(3): nat.gemm(A, W, C, M, N, K, K, K, N)
torch.cuda.synchronize(); print("ok", code, float(C.sum()))
