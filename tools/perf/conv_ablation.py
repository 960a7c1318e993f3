"""Where the implicit-GEMM convolution's time goes (csrc/conv_igemm.hip, 128 x 64 tiles): the encoder's widest layer forward (128 -> 256 channels, 42 x 42,
batch 256: 0.266 TFLOP) with parts of the kernel switched off (aid_conv_debug; results wrong by construction)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
from aid_amd import _native as nat
L = nat.lib()
shapes = [(128, 256, 256, 42, "encoder layer 3 forward"), (256, 128, 256, 42, "encoder layer 3 data gradient"), (512, 512, 16, 21, "ConvDecoder block 0 (16 images)"),
          (128, 256, 16, 84, "ConvDecoder block 3 main0 (16 images)"), (64, 64, 16, 168, "ConvDecoder block 3 main1 (16 images)"), (64, 32, 16, 168, "ConvDecoder out0 (16 images)")]
for Cin, Cout, B, H, name in shapes:
    K = 9 * Cin
    W = torch.randn(Cout, K, device="cuda") / K ** 0.5
    x = torch.randn(Cin, B, H, H, device="cuda")
    fl = 2.0 * Cout * K * B * H * H
    for mode, what in ((0, "full kernel"),) + (((4, "gather addresses fixed (no tap arithmetic), loads kept"), (1, "no global gathers"), (2, "no gathers, no B LDS writes"), (3, "no weight loads")) if os.environ.get("ABLATE") else ()):
        L.aid_conv_debug(mode)
        for _ in range(3): nat.conv3_igemm(W, x, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): nat.conv3_igemm(W, x, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:40s} {what:55s} {dt*1e3:7.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s", flush=True)
    L.aid_conv_debug(0)
    for vec in ((10, "TAP-MAJOR kernel, element gathers"), (11, "TAP-MAJOR kernel, 4-pixel vector gathers")) if Cin % 16 == 0 else ():
        L.aid_conv_debug(vec[0])
        for _ in range(3): nat.conv3_igemm_tm(W, x, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): nat.conv3_igemm_tm(W, x, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:40s} {vec[1] + ' (+ re-ordering launch)':55s} {dt*1e3:7.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s", flush=True)
    L.aid_conv_debug(11)
