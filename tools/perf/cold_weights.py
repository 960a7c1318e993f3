import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
st = torch.cuda.Stream()
def bench(fns):
    with torch.cuda.stream(st):
        for f in fns: f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for f in fns: f()
        g.replay(); torch.cuda.synchronize(); t0=time.time()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); return (time.time()-t0)/5/len(fns)
NW=48
for (M,N,K) in [(256,512,512),(512,512,512),(256,512,2048),(512,512,2048),(256,2048,512),(512,2048,512)]:
    Ws=[torch.randn(N,K,device="cuda") for _ in range(NW)]   # cycle through distinct weights: cold in L2 like the real loop
    bias=torch.randn(N,device="cuda")
    A=torch.randn(M,K,device="cuda"); C=torch.zeros(M,N,device="cuda")
    ref=(A.double()@Ws[0].double().t()+bias.double())
    row=[]
    t=bench([ (lambda w=w: nat.gemm(A,w,C,M,N,K,K,K,N,bias=bias)) for w in Ws]); row.append(f"store {t*1e6:6.1f}us")
    for ks in (1,2,4,8):
        if (K//256)%ks: continue
        C.zero_(); nat.gemm_ksplit_add(A,Ws[0],C,bias,ks); torch.cuda.synchronize()
        err=float((C.double()-ref).abs().max()/ref.abs().max())
        t=bench([ (lambda w=w: nat.gemm_ksplit_add(A,w,C,bias,ks)) for w in Ws]); row.append(f"ks{ks} {t*1e6:6.1f}us{'!' if err>1e-5 else ''}")
    print(f"M={M} N={N} K={K}: "+"  ".join(row))
