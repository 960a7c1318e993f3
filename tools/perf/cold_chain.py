"""One reverse-diffusion step's GEMM chain with distinct (L2-cold) weights per layer, graph-captured: which kernel variant is best IN SITU?
usage: cold_chain.py M"""
import sys, os, time, itertools
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
nat.lib()
M, D = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 512
kinds = ["attn", "mlp0", "mlp2"]
shape = {"attn": (D, D), "mlp0": (4 * D, D), "mlp2": (D, 4 * D)}
Ws = [(k, torch.randn(*shape[k], device="cuda") * 0.02) for _ in range(6) for k in kinds]
acts = {D: [torch.randn(M, D, device="cuda") for _ in range(2)], 4 * D: [torch.randn(M, 4 * D, device="cuda") for _ in range(2)]}
main = torch.cuda.Stream()
def step(force):
    for i, (kind, W) in enumerate(Ws):
        n, k = W.shape
        nat.gemm_force_kernel(force.get(kind, 0))
        nat.gemm(acts[k][i & 1], W, acts[n][(i + 1) & 1], M, n, k, k, k, n)
    nat.gemm_force_kernel(0)
def run(force, steps=10):
    with torch.cuda.stream(main):
        step(force); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            for _ in range(steps): step(force)
        g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / steps * 1e6
NAMES = {0: "auto", 1: "panel32", 2: "ring64", 4: "tile64", 6: "splitk", 9: "b64x32p2", 7: "b64x16p2", 10: "b64x16", 13: "b64x128x16", 14: "panel128", 15: "panel64"}
base = run({})
print(f"M={M}: auto chain {base:.1f} us per step ({base/18:.1f} us/GEMM)", flush=True)
for kind in kinds:
    row = []
    for code in (1, 14, 15, 2):
        try:
            t = run({kind: code})
            row.append(f"{NAMES[code]}={t - base:+.1f}")
        except Exception as e:
            row.append(f"{NAMES[code]}=ERR")
    print(f"  {kind:5s} (delta us/step vs auto, 6 launches): " + "  ".join(row), flush=True)
