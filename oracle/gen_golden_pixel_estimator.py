#!/usr/bin/env python3
"""Golden vectors for the pixel branch of FunctionSpaceEpistemicEstimator.forward (train mode for its own nets, decoder in eval) from a
run of the REFERENCE; container-only.  Asserts oracle.pixel_epistemic_value == reference; writes tests/golden/pixel_estimator.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aid_oracle as orc  # noqa: E402
from oracle import gen_golden as gg  # noqa: E402
from oracle.encoder_weights import (conv_decoder_specs, draw_conv_decoder_params, draw_pixel_estimator_params, pixel_estimator_specs)  # noqa: E402
from oracle.noise import NoiseStream  # noqa: E402


def main():
    torch.set_num_threads(8)
    gg.install_stubs(); sys.path.insert(0, gg.REF)
    tape = gg.Tape(); tape.install()
    from active_inference_diffusion.core.active_inference import FunctionSpaceEpistemicEstimator
    from active_inference_diffusion.encoder.visual_encoders import ConvDecoder
    L, B, S = 16, 3, 2
    KD, KE, KZ, K1, K2 = 5001, 5002, 5003, 5004, 5005
    tape.use(999)
    dec = ConvDecoder(latent_dim=L, output_dim=0, img_channels=3, hidden_dim=256, device=torch.device("cpu"))
    est = FunctionSpaceEpistemicEstimator(dec, L, (3, 84, 84), hidden_dim=64, spatial_aggregator_output_dim=256, is_pixel_observation=True, device="cpu")
    pd = draw_conv_decoder_params(conv_decoder_specs(L), KD)
    dec.load_state_dict(pd, strict=True)
    especs = pixel_estimator_specs(L)
    own = {k: v for k, v in est.state_dict().items() if not k.startswith("decoder.")}
    assert [(k, tuple(v.shape)) for k, v in own.items()] == especs, "pixel_estimator_specs differ from the reference state_dict"
    pe = draw_pixel_estimator_params(especs, KE)
    est.load_state_dict({**{"decoder." + k: v for k, v in pd.items()}, **pe}, strict=True)
    p = {"epistemic_estimator." + k: v.clone() for k, v in pe.items()}
    p.update({"epistemic_estimator.decoder." + k: v.clone() for k, v in pd.items()})
    zs = NoiseStream(KZ)
    nm = zs.randn(B, L); lv = np.full((B, L), np.log(0.1), np.float32)
    est.train()
    g = {"dims": np.array([L, B, S]), "keys": np.array([KD, KE, KZ, K1, K2])}
    rm = 0.0
    for call, key in ((1, K1), (2, K2)):
        tape.use(key)
        with torch.no_grad():
            epi_ref, met = est(torch.from_numpy(nm), torch.from_numpy(lv), num_samples=S)
        epi, mi, joint, tlog, rm = orc.pixel_epistemic_value(p, torch.from_numpy(nm), torch.from_numpy(lv), rm, NoiseStream(key), S, True)
        gg.close(np.float32(mi), np.float32(met["epistemic/mi_estimate"]), 2e-4, 2e-5, "pixel estimator mi")
        gg.close(np.float32(joint), np.float32(met["epistemic/joint_term"]), 2e-4, 2e-5, "pixel estimator joint")
        gg.close(np.float32(tlog), np.float32(met["epistemic/marginal_term"]), 2e-4, 2e-5, "pixel estimator marginal")
        gg.close(np.float32(rm), np.float32(met["epistemic/running_mean"]), 2e-4, 2e-5, "pixel estimator running mean")
        print(f"call {call}: reference", met, "oracle mi", float(mi))
        for k, v in met.items():
            g[f"c{call}/{k}"] = np.float64(v)
        g[f"c{call}/epistemic"] = epi_ref.numpy().copy()
    out = os.path.join(ROOT, "tests", "golden", "pixel_estimator.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
