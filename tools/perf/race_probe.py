"""Diagnostic: one pixel train_step under different schedules / repeated under the same one - which gradient and post-step tensors differ bit-wise?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "active-inference-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle.noise import NoiseStream
from parity_noise import ParityNoise
from test_hip_pixel import _pixel_agent, _load_pixel_params

L, A, D, N, H, B, cap = 32, 6, 64, int(os.environ.get("N", 400)), 1, int(os.environ.get("B", 160)), 192


def run(early, overlap=True):
    torch.manual_seed(0)          # the feature decoder keeps torch's default initialisation: same seed, same weights
    agent = _pixel_agent(L, A, D, N, H, B, cap=cap)
    ai = agent.active_inference
    _load_pixel_params(agent, L, A, D, N, 23)
    bs = NoiseStream(9)
    agent.replay_buffer.add_batch(bs.rand(cap, 9 * 84 * 84), np.tanh(bs.randn(cap, A)).astype(np.float32), bs.randn(cap), bs.rand(cap, 9 * 84 * 84), bs.rand(cap) < 0.1)
    pn = ParityNoise(555, pixel=True); ai.noise = pn; agent.replay_buffer.index_source = pn
    agent.early_belief_chains = early; agent.overlap_streams = overlap
    out = {}
    so = agent.score_optimizer
    orig = so.step

    def step():          # snapshot the gradients the step is about to consume (stream-ordered copies on the stepping stream)
        for name, ar in (("score", so.score), ("encoder", so.encoder), ("fdec", so.fdec)):
            for k, v in ar.grad_views().items():
                out["pre/" + name + "/" + k] = v.detach().clone()
        orig()
    so.step = step
    agent.train_step(); torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    for name, ar in list(ai.arenas.items()) + [("encoder", agent.encoder_arena)]:
        if ar.grad is None or ar.grad.numel() == 0:
            continue
        for k, v in ar.grad_views().items():
            out["grad/" + name + "/" + k] = v.detach().cpu().clone()
    sd = {**ai.state_dict(), **{"encoder." + k: v for k, v in agent.encoder.state_dict().items()}}
    out.update({"post/" + k: v.detach().cpu().clone() for k, v in sd.items()})
    out.update({"last/" + k: v.detach().cpu().clone() for k, v in agent._last.items()})
    del agent, ai; torch.cuda.empty_cache()
    return out


def cmp(a, b, tag):
    diff = [(k, float((a[k].double() - b[k].double()).abs().max()), float(a[k].double().abs().max()), int((a[k] != b[k]).sum()), a[k].numel(),
             float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-300))) for k in a if a[k].shape == b[k].shape and not torch.equal(a[k], b[k])]
    print(f"== {tag}: {len(diff)} of {len(a)} tensors differ")
    for k, d, m, n, tot, nw in [x for x in diff if not x[0].startswith("post/")][:60]:
        print(f"   {k}: max|diff| {d:.3e} (max|a| {m:.3e}), {n} of {tot} elements, norm-wise {nw:.3e}")


r1 = run(True); r2 = run(True); r3 = run(False); r5 = run(True, overlap=False)
cmp(r1, r2, "early twice"); cmp(r1, r3, "early vs paired"); cmp(r1, r5, "early vs single-stream")
