"""The estimator's full-row Linear -> LayerNorm -> SiLU kernels on the float32 MFMA pipe against the six-term bf16 split (aid_gemm_emulation(1),
csrc/gemm_rowln_x6.hip): time per launch and the error of BOTH against a float64 evaluation of the same layer."""
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
shapes = [(128000, 1024, 1024, True), (128000, 512, 1024, False), (128000, 512, 512, False), (25600, 1024, 1024, True), (25600, 512, 512, False)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")[:3]) + (a.split("x")[3] == "1",) for a in sys.argv[1:]]
for (M, N, K, skip) in shapes:
    torch.manual_seed(M + N + K)
    As = [torch.randn(M, K, device="cuda") for _ in range(3)]; W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); ga = torch.rand(N, device="cuda") + 0.5; be = torch.randn(N, device="cuda")
    add = torch.randn(M, N, device="cuda") if skip else None
    i = [0]
    def fused():
        i[0] = (i[0] + 1) % 3
        return nat.linear_ln_act(As[i[0]], W, b, ga, be, "silu", add)
    reps = 5 if M > 50000 else 15
    fl = 2.0 * M * N * K
    RR = 4096                                   # rows checked against float64
    x64 = As[0][:RR].double() @ W.double().t() + b.double()
    ref = torch.nn.functional.silu(torch.nn.functional.layer_norm(x64, (N,), ga.double(), be.double(), 1e-5)) + (add[:RR].double() if skip else 0)
    out = {}
    for mode in (0, 1):
        nat.lib().aid_gemm_emulation(mode)
        t = timeit(fused, reps)
        i[0] = 2; y = fused()
        e = (y[:RR].double() - ref).abs()
        out[mode] = (t, float(e.max()), float(e.pow(2).mean().sqrt()))
    nat.lib().aid_gemm_emulation(0)
    (t0, m0, r0), (t1, m1, r1) = out[0], out[1]
    print(f"M={M:6d} N={N:5d} K={K:5d}: f32 mfma {t0*1e6:8.1f} us ({fl/t0/1e12:6.1f} TF)  err max {m0:.2e} rms {r0:.2e} | bf16x6 {t1*1e6:8.1f} us ({fl/t1/1e12:6.1f} TF-eq)  err max {m1:.2e} rms {r1:.2e} | x{t0/t1:.2f}", flush=True)

# plain Linear (+ bias, ReLU) through aid_gemm_f32's tall route: the MINE statistics layers and the reverse loop's hoisted conditioning GEMM
for (M, N, K) in [(128000, 512, 384), (128000, 512, 512), (12800, 13312, 512)]:
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    RR = 2048
    ref = torch.relu(x[:RR].double() @ W.double().t() + b.double())
    out = {}
    for mode in (0, 1):
        nat.lib().aid_gemm_emulation(mode)
        t = timeit(lambda: nat.gemm(x, W, y, M, N, K, K, K, N, bias=b, act="relu"), 5)
        e = (y[:RR].double() - ref).abs()
        out[mode] = (t, float(e.max()), float(e.pow(2).mean().sqrt()))
    nat.lib().aid_gemm_emulation(0)
    (t0, m0, r0), (t1, m1, r1) = out[0], out[1]
    print(f"plain M={M:6d} N={N:5d} K={K:5d}: f32 mfma {t0*1e6:8.1f} us ({fl/t0/1e12:6.1f} TF)  err max {m0:.2e} rms {r0:.2e} | bf16x6 {t1*1e6:8.1f} us ({fl/t1/1e12:6.1f} TF-eq)  err max {m1:.2e} rms {r1:.2e} | x{t0/t1:.2f}", flush=True)
