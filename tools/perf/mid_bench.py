"""The reverse loop's GEMM shapes at 256 / 512 / 768 rows: the mid-M kernel (aid_gemm_mid) against the automatic dispatcher (aid_gemm_f32),
each in a chain that cycles through 6 distinct weight matrices (as in situ: a block's weights are cold every step), plus the whole
50-step loop through aid_denoise_loop."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for M in (256, 512, 768):
    for (N, K, ks, act) in ((512, 512, 4, "none"), (2048, 512, 1, "gelu"), (512, 2048, 4, "none")):
        NW = int(os.environ.get("MID_NW", "24"))
        A = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(NW)]   # 24 x 4 MB: beyond the 32 MB of L2
        b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda"); i = [0]
        def mid():
            i[0] = (i[0] + 1) % NW
            nat.gemm_mid(A, Ws[i[0]], b if ks == 1 else None, act, None, ks)
        def auto():
            i[0] = (i[0] + 1) % NW
            nat.gemm(A, Ws[i[0]], C, M, N, K, K, K, N, bias=b, act=act)
        tm, ta = timeit(mid, 96), timeit(auto, 96)
        fl = 2.0 * M * N * K
        print(f"M={M:4d} N={N:5d} K={K:5d} ksplit={ks}: gemm_mid {tm*1e6:6.1f} us ({fl/tm/1e12:5.1f} TF)   dispatcher {ta*1e6:6.1f} us ({fl/ta/1e12:5.1f} TF)", flush=True)
