"""Microbenchmark of the large-M GEMM kernel variants (force codes of aid_gemm_force_kernel; high byte = tile map)."""
import os, sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
NAMES = {0: "auto", 4: "tile64", 9: "big_pd2", 11: "big_pd1", 2: "ring64", 13: "big_bk64"}
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
shapes = [(25600, 1024, 1024), (25600, 512, 1024), (25600, 512, 512), (25600, 13312, 512), (25600, 512, 384), (2560, 512, 512), (12800, 512, 512), (25600, 256, 128)]
if len(sys.argv) > 1: shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    # a chain of distinct activations (as in situ: every GEMM reads a freshly produced A), weights re-used
    As = [torch.randn(M, K, device="cuda") for _ in range(3)]; W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    ref = None; line = f"M={M:6d} N={N:6d} K={K:5d}: "
    for code in ([int(c) for c in os.environ["BIG_GEMM_CODES"].split(",")] if os.environ.get("BIG_GEMM_CODES") else (4, 11, 11 + 256, 9, 9 + 256, 2, 0)):
        nat.gemm_force_kernel(code)
        i = [0]
        def fn():
            i[0] = (i[0] + 1) % 3
            nat.gemm(As[i[0]], W, C, M, N, K, K, K, N)
        dt = timeit(fn, 6 if M * N * K > 5e10 else 20)
        nat.gemm(As[0], W, C, M, N, K, K, K, N); torch.cuda.synchronize()
        if ref is None: ref = C.clone()
        err = float((C - ref).abs().max() / ref.abs().max())
        line += f"{NAMES.get(code & 255, str(code & 255))}{'/x' if code >> 8 else ''}={dt*1e6:7.1f}us({2.0*M*N*K/dt/1e12:5.1f}TF{'' if err < 1e-5 else ' ERR%.1e' % err}) "
    nat.gemm_force_kernel(0)
    print(line, flush=True)
