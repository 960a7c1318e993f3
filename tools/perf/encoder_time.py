import sys, os, time
R = "/root/repo"; sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd.encoder import DrQV2Encoder
B, L = 256, 128
enc = DrQV2Encoder((3, 84, 84), L, 3).to("cuda")
x = torch.rand(B, 9, 84, 84, device="cuda")
noise = lambda: {"shift_h": torch.randint(0, 9, (B,), device="cuda", dtype=torch.int32), "shift_w": torch.randint(0, 9, (B,), device="cuda", dtype=torch.int32)}
for it in range(3):
    f, ctx = enc(x, noise(), keep_for_backward=True); enc.backward(torch.randn_like(f), ctx)
torch.cuda.synchronize()
tf = tb = 0.0; n = 5
for it in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f, ctx = enc(x, noise(), keep_for_backward=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    enc.backward(torch.randn_like(f), ctx); torch.cuda.synchronize(); t2 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1
print(f"encoder B={B}: forward {1e3*tf/n:.2f} ms, backward {1e3*tb/n:.2f} ms, pair {1e3*(tf+tb)/n:.2f} ms")
