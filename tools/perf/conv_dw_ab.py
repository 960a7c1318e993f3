"""Weight gradient of the encoder's convolutions (csrc/conv_igemm.hip): the tap-major, vector-gather kernel against the element-gather one (aid_conv_debug 13 / 12)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
L = nat.lib()
for Cin, Cout, B, H, name in ((128, 256, 256, 42, "encoder layer 3"), (64, 128, 256, 42, "encoder layer 2")):
    K = 9 * Cin
    x = torch.randn(Cin, B, H, H, device="cuda"); dy = torch.randn(Cout, B, H, H, device="cuda")
    fl = 2.0 * Cout * K * B * H * H
    for mode, what in ((12, "element gathers, k = ci 9 + tap"), (14, "tap-major 64 x 64 tiles, 4-pixel vector gathers"), (15, "tap-major 128 x 128 tiles, prefetch distance 2")):
        L.aid_conv_debug(13); L.aid_conv_debug(15); L.aid_conv_debug(15); L.aid_conv_debug(mode)
        for _ in range(3): nat.conv3_igemm_dw(dy, x, K, 1, slabs=96)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): nat.conv3_igemm_dw(dy, x, K, 1, slabs=96)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:18s} {what:42s} {dt*1e3:7.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s", flush=True)
    L.aid_conv_debug(13); L.aid_conv_debug(15)
