"""AdamW + clip + EMA over a score-arena-sized buffer (27.15 M parameters): GB/s of the 40 B/parameter the three launches move."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
n = 27_150_000
p, g, m, v, e = (torch.randn(n, device="cuda") for _ in range(5)); v.abs_()
scratch = torch.zeros(8 + 4096, device="cuda")
def step(k):
    nat.check(nat.lib().aid_adamw_step(nat._ptr(p), nat._ptr(g), nat._ptr(m), nat._ptr(v), nat._ptr(e), nat.C.c_size_t(n), nat.C.c_float(5e-5), nat.C.c_float(1e-5),
                                       nat.C.c_float(0.9), nat.C.c_float(0.999), nat.C.c_float(1e-8), k, nat.C.c_float(0.5), nat.C.c_float(0.9999), nat._ptr(scratch), nat._stream()), "adamw")
for k in range(1, 4): step(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(4, 24): step(k)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"adamw step {dt*1e6:.1f} us  {n*40/dt/1e9:.0f} GB/s (40 B/param incl. the norm pass)")
