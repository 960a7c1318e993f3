"""Linear -> LayerNorm -> SiLU (+ skip): the full-row-tile kernel (aid_linear_ln_act, LayerNorm in the GEMM epilogue) against the
two-kernel path (64x64-tile GEMM, then torch's LayerNorm+SiLU as a stand-in for the row-wise kernel) on the estimator's shapes."""
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
shapes = [(25600, 1024, 1024, True), (25600, 512, 1024, False), (128000, 1024, 1024, True), (128000, 512, 1024, False), (25600, 512, 512, False)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")[:3]) + (a.split("x")[3] == "1",) for a in sys.argv[1:]]
for (M, N, K, skip) in shapes:
    As = [torch.randn(M, K, device="cuda") for _ in range(3)]; W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); ga = torch.rand(N, device="cuda") + 0.5; be = torch.randn(N, device="cuda")
    add = torch.randn(M, N, device="cuda") if skip else None
    C = torch.empty(M, N, device="cuda")
    i = [0]
    def fused():
        i[0] = (i[0] + 1) % 3
        return nat.linear_ln_act(As[i[0]], W, b, ga, be, "silu", add)
    def gemm_only():
        i[0] = (i[0] + 1) % 3
        nat.gemm(As[i[0]], W, C, M, N, K, K, K, N, bias=b)
    reps = 5 if M > 50000 else 15
    fl = 2.0 * M * N * K
    tf, tg = timeit(fused, reps), timeit(gemm_only, reps)
    i[0] = 2; y = fused(); i[0] = 2; gemm_only()         # both on As[0]
    ref = torch.nn.functional.silu(torch.nn.functional.layer_norm(C, (N,), ga, be, 1e-5)) + (add if skip else 0)
    err = float((y - ref).abs().max())
    extra = ""
    if os.environ.get("ROWLN_AB"):
        nat.lib().aid_rowln_debug(16); t = timeit(fused, reps); nat.lib().aid_rowln_debug(17)
        extra += f"  register-staged {t*1e6:8.1f} us ({fl/t/1e12:5.1f} TF) "
    if os.environ.get("ROWLN_DBG"):
        for mode in (1, 2, 4, 3):
            nat.lib().aid_rowln_debug(mode); t = timeit(fused, reps); nat.lib().aid_rowln_debug(0)
            extra += f"  dbg{mode} {t*1e6:8.1f} us ({fl/t/1e12:5.1f} TF)"
    print(f"M={M:6d} N={N:5d} K={K:5d}:{extra} fused {tf*1e6:8.1f} us ({fl/tf/1e12:5.1f} TF)   plain GEMM alone {tg*1e6:8.1f} us ({fl/tg/1e12:5.1f} TF)   max|diff| {err:.2e}", flush=True)
