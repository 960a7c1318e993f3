import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
st = torch.cuda.Stream()
def bench(label, fn, reps=200, flops=0):
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        t0=time.time()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); dt=(time.time()-t0)/5/reps
    print(f"{label:40s} {dt*1e6:8.2f} us/kernel" + (f"  {flops/dt/1e12:7.2f} TF/s" if flops else ""))
x4 = torch.empty(4, device="cuda")
bench("trivial (philox n=4)", lambda: nat.fill_normal(x4, 1, 0))
xl = torch.empty(256*512, device="cuda")
bench("philox n=131072", lambda: nat.fill_normal(xl, 1, 0))
for (M,N,K) in [(256,512,512),(512,512,512),(256,2048,512),(256,512,2048),(512,2048,512),(512,512,2048),(256,1024,512),(256,13312,512),(512,13312,512),(2560,512,512),(25600,512,512),(4096,4096,4096)]:
    A=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); C=torch.empty(M,N,device="cuda")
    bench(f"gemm nt M={M} N={N} K={K}", lambda: nat.gemm(A,W,C,M,N,K,K,K,N), reps=100 if M*N*K<1e10 else 10, flops=2.0*M*N*K)
