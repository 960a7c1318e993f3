"""Per-GEMM timings (library profile hooks) of one DrQ-v2 encoder forward and one backward at B frame stacks: which conv GEMM shapes cost what."""
import sys, os, csv, collections, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
from aid_amd.encoder import DrQV2Encoder
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 128
enc = DrQV2Encoder((3, 84, 84), L, 3).to("cuda")
x = torch.rand(B, 9, 84, 84, device="cuda")
noise = lambda: {"shift_h": torch.randint(0, 9, (B,), device="cuda", dtype=torch.int32), "shift_w": torch.randint(0, 9, (B,), device="cuda", dtype=torch.int32)}
for it in range(2):
    f, ctx = enc(x, noise(), keep_for_backward=True); enc.backward(torch.randn_like(f), ctx)
torch.cuda.synchronize()
for phase in ("forward", "backward"):
    nat.prof_enable(True); nat.prof_reset()
    if phase == "forward": f, ctx = enc(x, noise(), keep_for_backward=True)
    else: enc.backward(torch.randn_like(f), ctx)
    torch.cuda.synchronize()
    path = tempfile.mktemp(suffix=".csv"); nat.prof_dump(path); nat.prof_enable(False); nat.prof_reset()
    rows = [r for r in csv.DictReader(open(path)) if int(r["kernel_id"]) in (0, 3)]
    tot = sum(float(r["ms"]) for r in rows)
    print(f"{phase}: {len(rows)} GEMM launches, {tot:.2f} ms")
    for r in sorted(rows, key=lambda r: -float(r["ms"]))[:12]:
        M, N, K = int(r["M"]), int(r["N"]), int(r["K"]); ms = float(r["ms"])
        print(f"   variant {r['variant']:>3} layout {r['layout']:>2}  M={M:7d} N={N:7d} K={K:7d}  {ms:7.3f} ms  {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s")
