import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
import aid_amd
from aid_amd.encoder import DrQV2Encoder
from aid_amd.arena import ParamArena
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 128
enc = DrQV2Encoder((3, 84, 84), L, 3).to("cuda")
arena = ParamArena([("encoder." + n, p) for n, p in enc.trained_parameters()], "cuda")
x = torch.rand(B, 9, 84, 84, device="cuda")
noise = lambda: {"shift_h": torch.randint(0, 9, (B,), device="cuda", dtype=torch.int32), "shift_w": torch.randint(0, 9, (B,), device="cuda", dtype=torch.int32)}
for it in range(3):
    f, ctx = enc(x, noise(), keep_for_backward=True); enc.backward(torch.randn_like(f), ctx)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(5): f, ctx = enc(x, noise(), keep_for_backward=True)
torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 5
d = torch.randn_like(f); t0 = time.perf_counter()
for it in range(5): enc.backward(d, ctx)
torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 5
print(f"B={B} L={L}: encoder fwd {tf*1e3:.2f} ms, bwd {tb*1e3:.2f} ms")
