"""GEMMs of the reverse loop at act() / collector batch sizes (M = 1, 8, 32): automatic dispatcher vs the panel kernel's atomic split-K.
The same launch repeated inside one hipGraph (weights L2-warm) and a 6-block chain over distinct weights (weights from the Infinity Cache)."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
st = torch.cuda.Stream()
def graph_time(fn, reps):
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        t0=time.time()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); return (time.time()-t0)/5/reps
codes = [int(x) for x in sys.argv[1:]] or [0]
for M in (1, 8, 32):
    for (N, K) in ((512, 512), (2048, 512), (512, 2048)):
        A = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") for _ in range(24)]; C = torch.zeros(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
        row = []
        for code in codes:
            nat.gemm_force_kernel(code)
            def chain():
                for W in Ws: nat.gemm(A, W, C, M, N, K, K, K, N, bias=bias)
            try:
                row.append(f"code{code}: warm {graph_time(lambda: nat.gemm(A, Ws[0], C, M, N, K, K, K, N, bias=bias), 50)*1e6:5.2f} us, chain {graph_time(chain, 4)/len(Ws)*1e6:5.2f} us")
            except Exception as e:
                row.append(f"code{code}: ERR {str(e)[:40]}")
            nat.gemm_force_kernel(0)
        for ks in (2, 4, 8):
            if K // ks < 256: continue
            row.append(f"ksplit{ks}: warm {graph_time(lambda: nat.gemm_ksplit_add(A, Ws[0], C, bias, ks), 50)*1e6:5.2f} us")
        print(f"M={M:3d} N={N:5d} K={K:5d}: " + " | ".join(row))
