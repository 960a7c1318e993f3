import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, numpy as np
from aid_amd import _native as nat
nat.lib()
st = torch.cuda.Stream()
def bench(fn, reps=50):
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        t0=time.time()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); return (time.time()-t0)/3/reps
names={1:"panel32",14:"panel128",15:"panel64",2:"ring64",4:"tile64",9:"big",6:"splitk"}   # aid_gemm_force_kernel codes (include/aid_hip.h)
shapes=[(256,512,512,"nt"),(512,512,512,"nt"),(256,2048,512,"nt"),(512,2048,512,"nt"),(256,512,2048,"nt"),(512,512,2048,"nt"),(256,13312,512,"nt"),(512,13312,512,"nt"),
        (2560,512,512,"nt"),(2560,256,512,"nt"),(25600,1024,1024,"nt"),(25600,512,1024,"nt"),(25600,512,384,"nt"),(25600,256,128,"nt"),
        (256,512,2048,"nn"),(512,2048,256,"tn"),(13312,512,256,"tn"),(2560,512,512,"nn"),(512,512,2560,"tn")]
for (M,N,K,lay) in shapes:
    akm = 1 if lay=="tn" else 0; bkm = 0 if lay=="nt" else 1
    A=torch.randn((K,M) if akm else (M,K),device="cuda"); W=torch.randn((K,N) if bkm else (N,K),device="cuda"); C=torch.empty(M,N,device="cuda")
    ref = (A.t() if akm else A).double() @ (W if bkm else W.t()).double()
    row=[]
    for k in names:
        nat.gemm_force_kernel(k)
        try:
            C.zero_(); nat.gemm(A,W,C,M,N,K,A.shape[1],W.shape[1],N,akm,bkm); torch.cuda.synchronize()
            err=float((C.double()-ref).abs().max()/ref.abs().max())
            dt=bench(lambda: nat.gemm(A,W,C,M,N,K,A.shape[1],W.shape[1],N,akm,bkm), reps=50 if M*N*K<2e10 else 10)
            row.append(f"{names[k]}={dt*1e6:7.1f}us({2.0*M*N*K/dt/1e12:5.1f}TF{'!' if err>1e-5 else ''})")
        except Exception as e:
            row.append(f"{names[k]}=ERR")
    nat.gemm_force_kernel(0)
    dt=bench(lambda: nat.gemm(A,W,C,M,N,K,A.shape[1],W.shape[1],N,akm,bkm), reps=50 if M*N*K<2e10 else 10)
    print(f"{lay} M={M:6d} N={N:6d} K={K:5d}: "+" ".join(row)+f" | auto={dt*1e6:7.1f}us")
