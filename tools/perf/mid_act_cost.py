"""What the GELU epilogue costs the reverse loop's 256 x 2048 x 512 product (kernel durations from `rocprofv3 --kernel-trace --stats -- python3
tools/perf/mid_act_cost.py <none|gelu|down>`): the same shape with and without the activation, and the 512 x 2048 down projection in 4 K slices."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
what = sys.argv[1]
M = 256
N, K, ks, act = {"none": (2048, 512, 1, "none"), "gelu": (2048, 512, 1, "gelu"), "down": (512, 2048, 4, "none"), "sq1": (512, 512, 1, "none"), "sq4": (512, 512, 4, "none"), "sq2": (512, 512, 2, "none")}[what]
NW = int(os.environ.get("MID_NW", "24"))          # 24 x 4 MB of weights: cold in the L2s every launch, as in situ; MID_NW=1: warm
A = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(NW)]; b = torch.randn(N, device="cuda")
for i in range(240): nat.gemm_mid(A, Ws[i % NW], b if ks == 1 else None, act, None, ks)
torch.cuda.synchronize()
