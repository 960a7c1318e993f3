"""Tile shapes of the tap-major convolution kernel on the big layers: 128 x 128 tiles (a wave owns 128 x 32; what >= 128 output channels over >= 65 536 pixels take) against
128 x 64 (aid_conv_debug(22) forces it).  The 32-deep-stage form of the 128 x 64 tile measured 97-104 TFLOP/s where the 16-deep one has 105-112 and was dropped."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
L = nat.lib()
for Cin, Cout, B, H, name in ((128, 256, 256, 42, "encoder layer 3 forward"), (256, 128, 256, 42, "encoder layer 3 data gradient"), (64, 128, 256, 42, "encoder layer 2 forward"),
                              (512, 512, 16, 21, "ConvDecoder block 0"), (128, 256, 16, 84, "ConvDecoder block 3 main0")):
    K = 9 * Cin
    W = torch.randn(Cout, K, device="cuda") / K ** 0.5
    x = torch.randn(Cin, B, H, H, device="cuda")
    fl = 2.0 * Cout * K * B * H * H
    ref = None
    for mode, what in ((22, "128 x 64, 16-deep"), (0, "automatic (128 x 128 on big planes)")):
        L.aid_conv_debug(mode)
        out = nat.conv3_igemm_tm(W, x, 1)
        if ref is None: ref = out
        else: assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4), (name, what, float((out - ref).abs().max()))
        for _ in range(3): nat.conv3_igemm_tm(W, x, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): nat.conv3_igemm_tm(W, x, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:32s} {what:28s} {dt*1e3:7.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s", flush=True)
    L.aid_conv_debug(0)
