#!/usr/bin/env python3
"""Golden vectors for ConvDecoder.forward (eval mode) from the REFERENCE; container-only (stubs / tape of gen_golden.py).
Asserts oracle.conv_decoder_forward == reference, writes tests/golden/conv_decoder.npz (inputs + expected images are regenerated /
stored as a strided sample + checksums: a 168x168x3 image per latent)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aid_oracle as orc  # noqa: E402
from oracle import gen_golden as gg  # noqa: E402
from oracle.encoder_weights import conv_decoder_specs, draw_conv_decoder_params  # noqa: E402
from oracle.noise import NoiseStream  # noqa: E402


def main():
    torch.set_num_threads(8)
    gg.install_stubs(); sys.path.insert(0, gg.REF)
    from active_inference_diffusion.encoder.visual_encoders import ConvDecoder
    L, R, KW, KZ = 16, 3, 4001, 4002
    dec = ConvDecoder(latent_dim=L, output_dim=0, img_channels=3, hidden_dim=256, device=torch.device("cpu"))
    specs = conv_decoder_specs(L)
    sd = dec.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == specs, "conv_decoder_specs differ from the reference state_dict"
    p = draw_conv_decoder_params(specs, KW)
    dec.load_state_dict(p, strict=True); dec.eval()
    z = NoiseStream(KZ).randn(R, L)
    with torch.no_grad():
        y_ref = dec(torch.from_numpy(z))
        y = orc.conv_decoder_forward({"observation_decoder." + k: v for k, v in p.items()}, torch.from_numpy(z))
    err = gg.close(y, y_ref, 1e-5, 1e-6, "conv decoder forward")
    print("oracle == reference, max abs err", err, "output", tuple(y_ref.shape), "range", float(y_ref.min()), float(y_ref.max()), "std", float(y_ref.std()))
    g = {"dims": np.array([L, R]), "keys": np.array([KW, KZ]), "sample": y_ref.reshape(-1)[::61].numpy().copy(),
         "checksum": np.array([float(y_ref[i].double().sum()) for i in range(R)]), "abssum": np.array([float(y_ref[i].double().abs().sum()) for i in range(R)]),
         "row": y_ref[1, :, 80, :].numpy().copy()}
    out = os.path.join(ROOT, "tests", "golden", "conv_decoder.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
