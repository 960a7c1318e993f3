"""A few launches of the fused Linear->LayerNorm->SiLU kernel on one shape (for rocprofv3 --pmc passes)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
if len(sys.argv) > 5: nat.lib().aid_rowln_debug(int(sys.argv[5]))      # 1-4: ablation variants, 16: register staging instead of LDS-DMA
As = [torch.randn(M, K, device="cuda") for _ in range(2)]; W = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda"); ga = torch.rand(N, device="cuda") + 0.5; be = torch.randn(N, device="cuda")
for i in range(reps):
    y = nat.linear_ln_act(As[i & 1], W, b, ga, be, "silu", None)
torch.cuda.synchronize()
print("done", float(y[0, 0]))
