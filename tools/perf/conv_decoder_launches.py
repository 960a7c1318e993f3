"""Per-GEMM timings (library profile hooks) of one ConvDecoder forward on R images (hidden 512, latent 128): SURVEY 8(f)-1."""
import sys, os, tempfile
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R_ + "/active-inference-diffusion_amd"); sys.path.insert(0, R_)
import torch, time, bench, aid_amd
from aid_amd import _native as nat
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=64, efe_horizon=1, latent_dim=128, hidden_dim=512, pixel_observation=True)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=4)
class Env4: observation_space, action_space = bench._Box((3, 3, 84, 84)), bench._Box((6,))
agent = aid_amd.DiffusionPixelAgent(Env4(), cfg, aid_amd.TrainingConfig(buffer_size=64), aid_amd.PixelObservationConfig())
ai = agent.active_inference
dec = ai.observation_decoder._tab.get(ai.observation_decoder)
z = torch.randn(R, 128, device="cuda")
for _ in range(2): nat.conv_decoder_forward(dec, z, ws=ai._ws)
torch.cuda.synchronize(); t0 = time.perf_counter(); nat.conv_decoder_forward(dec, z, ws=ai._ws); torch.cuda.synchronize(); print(f"{R} images: {1e3*(time.perf_counter()-t0):.2f} ms")
nat.prof_enable(True); nat.prof_reset(); nat.conv_decoder_forward(dec, z, ws=ai._ws); torch.cuda.synchronize()
path = tempfile.mktemp(suffix=".csv"); nat.prof_dump(path); nat.prof_enable(False)
os.system(f"python {R_}/tools/perf/launch_table.py {path} | head -24")
