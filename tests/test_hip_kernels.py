"""GPU parity: HIP kernels (through the C ABI) against the oracle / float64 numpy on the same inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import aid_oracle as orc
from oracle.noise import NoiseStream
from oracle.weights import draw_params, param_specs


@pytest.fixture(scope="module")
def nat():
    from aid_amd import _native
    _native.lib()
    return _native


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda().contiguous()


@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (256, 2048, 512), (256, 512, 2048), (8, 17, 64), (256, 13312, 512),
                                   (2560, 512, 384), (33, 70, 145), (1, 1024, 128), (300, 1, 512), (4096, 1024, 1024)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_layouts(nat, M, N, K, layout):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    akm = 1 if layout == "tn" else 0
    bkm = 0 if layout == "nt" else 1
    Ad = dev(A.T if akm else A); Bd = dev(Bm.T if bkm else Bm)
    out = torch.zeros(M, N, device="cuda")
    nat.gemm(Ad, Bd, out, M, N, K, Ad.shape[1], Bd.shape[1], N, akm, bkm, bias=dev(bias))
    got = out.cpu().numpy().astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64).T + 1.0
    assert np.max(np.abs(got - (ref + bias)) / scale) < 2e-6


@pytest.mark.parametrize("act", ["relu", "silu", "gelu"])
def test_gemm_epilogues(nat, act):
    rng = np.random.default_rng(5)
    M, N, K = 96, 200, 72
    A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((N, K)).astype(np.float32) / 8
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32)
    x = torch.from_numpy(A) @ torch.from_numpy(W).T + torch.from_numpy(b)
    ref = {"relu": torch.relu, "silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu}[act](x) + torch.from_numpy(R)
    out = torch.zeros(M, N, device="cuda")
    nat.gemm(dev(A), dev(W), out, M, N, K, K, K, N, bias=dev(b), act=act, residual=dev(R), ldr=N)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    # accumulate + alpha
    base = rng.standard_normal((M, N)).astype(np.float32)
    out = dev(base)
    nat.gemm(dev(A), dev(W), out, M, N, K, K, K, N, alpha=0.5, accumulate=1)
    np.testing.assert_allclose(out.cpu().numpy(), base + 0.5 * (A @ W.T), rtol=1e-5, atol=1e-5)


def _golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"state_{tag}.npz"))
    O, A, L, D, N = (int(g[k]) for k in ("O", "A", "L", "D", "N"))
    specs = param_specs(O, A, L, D, N)
    p = draw_params(specs, int(g["weights_key"]), orc.schedule_tables(N))
    return g, p, {k: v.cuda().contiguous() for k, v in p.items()}


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_score_forward_golden(nat, golden_dir, tag):
    """All four time branches of the reference forward (golden = reference outputs), eval + train."""
    g, p, pd = _golden(golden_dir, tag)
    L, D, B = int(g["L"]), int(g["D"]), int(g["B"])
    tab = nat.score_net_table(pd, "latent_score_network.", L, L, D)
    z, obs = dev(g["z"]), dev(g["obs"])
    for nm, tt, cont in (("cont", g["t_cont"], True), ("disc", np.full(B, 3.0), False), ("t0", np.zeros(B), True), ("t1", np.ones(B), True)):
        out = nat.score_forward(tab, z, dev(tt), obs, cont)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"score_{nm}"], rtol=1e-4, atol=1e-6, err_msg=nm)
    mask = NoiseStream(int(g["score_train_key"])).dropout_mask((B, D), 0.1)
    out = nat.score_forward(tab, z, dev(g["t_cont"]), obs, True, drop_mask=dev(mask))
    np.testing.assert_allclose(out.cpu().numpy(), g["score_train"], rtol=1e-4, atol=1e-6)


def _loop_noise(key, B, L, D, N, train):
    ns = NoiseStream(key)
    z0 = ns.randn(B, L)
    eps = np.zeros((N, B, L), np.float32); masks = np.ones((N, B, D), np.float32)
    for t in reversed(range(N)):
        if train:
            masks[t] = ns.dropout_mask((B, D), 0.1)
        if t > 0:
            eps[t] = ns.randn(B, L)
    return z0, eps, masks


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_denoise_loop_golden(nat, golden_dir, tag):
    """Reverse-diffusion trajectory against the reference's own trajectory (train + eval mode)."""
    g, p, pd = _golden(golden_dir, tag)
    L, D, B, N = int(g["L"]), int(g["D"]), int(g["B"]), int(g["N"])
    tab = nat.score_net_table(pd, "latent_score_network.", L, L, D)
    coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
    z0, eps, masks = _loop_noise(int(g["belief_key"]), B, L, D, N, True)
    out, traj = nat.denoise_loop(tab, coefs, dev(g["obs"]), N, dev(z0), dev(eps), dev(masks), train=True, keep_traj=True)
    np.testing.assert_allclose(traj.cpu().numpy(), g["belief_traj"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), g["belief_latent"], rtol=1e-3, atol=1e-5)
    z0, eps, _ = _loop_noise(int(g["belief_eval_key"]), B, L, D, N, False)
    out = nat.denoise_loop(tab, coefs, dev(g["obs"]), N, dev(z0), dev(eps), None, train=False)
    np.testing.assert_allclose(out.cpu().numpy(), g["belief_eval_latent"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("O,A,B,N", [(17, 6, 256, 50), (11, 3, 1024, 10), (11, 3, 3072, 10), (17, 6, 1536, 10)])
def test_score_and_loop_full_size_vs_oracle(nat, O, A, B, N):
    """BASELINE configs 2/3 shapes (D=512): HIP vs the oracle on the same seeded inputs."""
    D, L = 512, O
    specs = param_specs(O, A, L, D, N)
    p = draw_params(specs, 4242 + O, orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    tab = nat.score_net_table(pd, "latent_score_network.", L, L, D)
    ns = NoiseStream(99)
    z, obs, t = ns.randn(B, L), ns.randn(B, O), ns.rand(B)
    T = torch.from_numpy
    ref = orc.score_forward(p, T(z), T(t), T(obs))
    out = nat.score_forward(tab, dev(z), dev(t), dev(obs), True)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-6)
    ref = orc.score_forward(p, T(z), torch.full((B,), 7.0), T(obs))
    out = nat.score_forward(tab, dev(z), dev(np.full(B, 7.0)), dev(obs), False)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-6)
    if B <= 256 or B > 1024:      # 3 072 / 1 536 rows: the merged loops of a batch-1024 / batch-512 step (128-row tiles, 1 / 2 K slices)
        nsteps = 10 if B <= 256 else 3  # oracle on CPU: keep it to seconds
        tabs = orc.schedule_tables(nsteps)
        coefs = orc.p_sample_coefs(tabs).cuda().contiguous()
        z0, eps, masks = _loop_noise(123, B, L, D, nsteps, True)
        ref = orc.generate_latent_trajectory(p, tabs, T(obs), NoiseStream(123), True)
        out = nat.denoise_loop(tab, coefs, dev(obs), nsteps, dev(z0), dev(eps), dev(masks), train=True)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-5)


def test_philox_fills_statistics(nat):
    n = 1 << 20
    x = torch.empty(n, device="cuda")
    nat.fill_normal(x, 7, 0)
    assert abs(float(x.mean())) < 5e-3 and abs(float(x.std()) - 1) < 5e-3
    y = torch.empty(n, device="cuda"); nat.fill_normal(y, 7, 0)
    assert torch.equal(x, y)  # counter-based: same (seed, offset) -> same stream
    nat.fill_normal(y, 7, n // 4)
    assert not torch.equal(x, y)
    nat.fill_uniform(x, 3, 0)
    assert 0 <= float(x.min()) and float(x.max()) < 1 and abs(float(x.mean()) - 0.5) < 2e-3
    nat.fill_dropout_mask(x, 0.1, 3, 0)
    assert abs(float((x == 0).float().mean()) - 0.1) < 2e-3 and abs(float(x.max()) - 1 / 0.9) < 1e-6


def test_errors_are_reported_not_thrown(nat):
    import ctypes as C
    rc = nat.lib().aid_gemm_f32(None, 1, 0, None, 1, 0, None, 1, 4, 4, 4, None, 0, None, 0, C.c_float(1.0), 0, None)
    assert rc == -1 and b"null" in nat.lib().aid_last_error()


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_collector_sampling_golden(nat, golden_dir, tag):
    """A6: AsyncGPUInference._batch_diffusion_sampling (reference output in tests/golden) + the NaN/Inf re-initialisation."""
    g, p, pd = _golden(golden_dir, tag)
    L, D, B, N = int(g["L"]), int(g["D"]), int(g["B"]), int(g["N"])
    tab = nat.score_net_table(pd, "latent_score_network.", L, L, D)
    coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
    steps = int(g["collector_steps"])
    ns = NoiseStream(int(g["collector_key"]))
    z0 = ns.randn(B, L); eps = np.zeros((steps, B, L), np.float32)
    for s_ in reversed(range(steps)):
        if s_ > 0:
            eps[s_] = ns.randn(B, L)
    out, cnt = nat.collector_sampling(tab, coefs, dev(g["obs"]), steps, N - 1, dev(z0), dev(eps), None, dev(np.zeros((steps, B, L), np.float32)), train=False)
    np.testing.assert_allclose(out.cpu().numpy(), g["collector_z"], rtol=1e-3, atol=1e-5)
    assert int(cnt) == 0
    # a NaN in the start state poisons the step-(steps-1) iterate -> whole batch re-initialised to 0.1*noise at that step
    z_bad = z0.copy(); z_bad[0, 0] = np.nan
    rn = np.random.default_rng(0).standard_normal((steps, B, L)).astype(np.float32)
    out2, cnt2 = nat.collector_sampling(tab, coefs, dev(g["obs"]), steps, N - 1, dev(z_bad), dev(eps), None, dev(rn), train=False)
    assert int(cnt2) == 1 and np.all(np.isfinite(out2.cpu().numpy()))
    ref = orc.collector_sampling(p, orc.schedule_tables(N), torch.from_numpy(g["obs"]),
                                 type("R", (), {"randn": staticmethod(lambda *s, _it=iter([z_bad, eps[2], rn[2], eps[1]]): next(_it))})(), steps, N, False)
    np.testing.assert_allclose(out2.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(512, 512, 17), (256, 512, 11), (100, 70, 3), (2560, 128, 17), (64, 64, 32), (1, 1, 1)])
def test_gemm_small_k_route(nat, M, N, K):
    """K <= 32 problems go to the VALU small-K kernel: same result as the fp64 product, with the fused epilogue."""
    ns = NoiseStream(77)
    A, W, b, res = ns.randn(M, K), ns.randn(N, K), ns.randn(N), ns.randn(M, N)
    C = torch.empty(M, N, device="cuda")
    nat.gemm(dev(A), dev(W), C, M, N, K, K, K, N, bias=dev(b), act="silu", residual=dev(res), ldr=N)
    pre = A.astype(np.float64) @ W.astype(np.float64).T + b
    ref = pre / (1 + np.exp(-pre)) + res
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("code", [1, 2, 4, 6, 9, 9 + 256, 11, 11 + 256, 14, 15, 512])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_every_kernel_variant(nat, code, layout):
    """Each kernel of the family (forced through aid_gemm_force_kernel) on a shape all of them accept, incl. a ragged row count for the
    k-contiguous-A layouts; same fp64 reference as the automatic dispatcher's tests."""
    M, N, K = (768, 512, 1024) if layout == "tn" else (700, 512, 1024)
    rng = np.random.default_rng(code)
    A = rng.standard_normal((M, K)).astype(np.float32); Bm = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    akm = 1 if layout == "tn" else 0; bkm = 0 if layout == "nt" else 1
    Ad = dev(A.T if akm else A); Bd = dev(Bm.T if bkm else Bm)
    out = torch.zeros(M, N, device="cuda")
    nat.gemm_force_kernel(code)
    try:
        nat.gemm(Ad, Bd, out, M, N, K, Ad.shape[1], Bd.shape[1], N, akm, bkm, bias=dev(bias))
    finally:
        nat.gemm_force_kernel(0)
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T + bias
    scale = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64).T + 1.0
    assert np.max(np.abs(out.cpu().numpy() - ref) / scale) < 2e-6


@pytest.mark.parametrize("M,N", [(64 * 37, 64 * 8), (64 * 13 + 5, 64 * 24), (64 * 9, 64 * 3), (64 * 41 + 63, 64 * 16 + 1)])
def test_gemm_big_kernel_xcd_grouped_map_any_grid(nat, M, N):
    """The XCD-grouped tile map of gemm_big_kernel must be a bijection for every grid (row-tile counts that are not multiples of 8,
    leftover tasks, column-tile counts below / not divisible by 8): forced kernel 9 with the map bit, against the fp64 reference."""
    K = 96
    rng = np.random.default_rng(M + N)
    A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((N, K)).astype(np.float32)
    out = torch.full((M, N), float("nan"), device="cuda")
    nat.gemm_force_kernel(9 + 256)
    try:
        nat.gemm(dev(A), dev(W), out, M, N, K, K, K, N)
    finally:
        nat.gemm_force_kernel(0)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    scale = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1.0
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.max(np.abs(got - ref) / scale) < 2e-6


@pytest.mark.parametrize("M,N,K,acc", [(6, 256, 2560, 1), (512, 17, 2560, 1), (17, 23, 1001, 0), (3, 1, 513, 1)])
def test_gemm_thin_weight_gradient(nat, M, N, K, acc):
    """dW[M,N] (+)= dY^T X over K rows with a 6- / 17-wide side: the automatic dispatcher's thin TN kernel (16 waves split K inside one
    workgroup, ordered fold) and the same kernel forced (code 8); ragged M, N and K; fp64 reference; two launches are bit-identical."""
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((K, M)).astype(np.float32); Bm = rng.standard_normal((K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64).T @ Bm.astype(np.float64) + (C0 if acc else 0.0)
    scale = np.abs(A).astype(np.float64).T @ np.abs(Bm).astype(np.float64) + 1.0
    outs = []
    for code in (0, 8, 8):
        out = dev(C0.copy())
        nat.gemm_force_kernel(code)
        try:
            nat.gemm(dev(A), dev(Bm), out, M, N, K, M, N, N, 1, 1, accumulate=acc)
        finally:
            nat.gemm_force_kernel(0)
        outs.append(out.cpu().numpy())
        assert np.max(np.abs(outs[-1] - ref) / scale) < 2e-6, code
    assert np.array_equal(outs[1], outs[2])


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (8, 2048, 512), (32, 512, 2048), (5, 17, 64), (27, 130, 1000), (3, 6, 4096)])
def test_gemm_skinny_rows(nat, M, N, K):
    """act() / collector batches (M <= 32): the wave-per-column weight-streaming kernel the dispatcher picks, and the same kernel forced
    (code 10), with bias + SiLU + residual; ragged M, N, K; fp64 reference; two launches bit-identical."""
    rng = np.random.default_rng(M + N)
    A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32); res = rng.standard_normal((M, N)).astype(np.float32)
    pre = A.astype(np.float64) @ W.astype(np.float64).T + bias
    ref = pre / (1 + np.exp(-pre)) + res
    scale = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1.0
    outs = []
    for code in (0, 10, 10):
        out = torch.zeros(M, N, device="cuda")
        nat.gemm_force_kernel(code)
        try:
            nat.gemm(dev(A), dev(W), out, M, N, K, K, K, N, bias=dev(bias), act="silu", residual=dev(res), ldr=N)
        finally:
            nat.gemm_force_kernel(0)
        outs.append(out.cpu().numpy())
        assert np.max(np.abs(outs[-1] - ref) / scale) < 2e-6, code
    assert np.array_equal(outs[1], outs[2]) and np.array_equal(outs[0], outs[1])


def test_replay_buffer_uint8_frames(nat):
    """optimize_memory=True + uint8 frames: the ring arrays stay bytes in HBM; sample() returns the same float values a float32 store
    would (the reference's arrays hold float32(frame)); bit-exact."""
    from aid_amd.utils.buffers import ReplayBuffer
    rng = np.random.default_rng(3)
    cap, shape, A, B = 32, (3, 3, 84, 84), 4, 16
    frames = rng.integers(0, 256, (cap, *shape), dtype=np.uint8); nxt = rng.integers(0, 256, (cap, *shape), dtype=np.uint8)
    act = rng.uniform(-1, 1, (cap, A)).astype(np.float32); rew = rng.standard_normal(cap).astype(np.float32)
    rb8 = ReplayBuffer(cap, shape, A, "cuda", optimize_memory=True); rbf = ReplayBuffer(cap, shape, A, "cuda", optimize_memory=False)
    for i in range(cap):
        rb8.add(frames[i], act[i], float(rew[i]), nxt[i], i % 7 == 0); rbf.add(frames[i], act[i], float(rew[i]), nxt[i], i % 7 == 0)
    assert rb8.observations.dtype == torch.uint8 and rbf.observations.dtype == torch.float32
    assert rb8.observations.element_size() * rb8.observations.numel() * 4 == rbf.observations.element_size() * rbf.observations.numel()

    class Fixed:
        def replay_indices(self, size, batch): return np.arange(batch)[::-1] % size
    rb8.index_source = rbf.index_source = Fixed()
    a, b = rb8.sample(B), rbf.sample(B)
    for k in a:
        assert a[k].dtype == torch.float32 and torch.equal(a[k], b[k]), k
    assert torch.equal(a["observations"][0].cpu(), torch.from_numpy(frames[B - 1].astype(np.float32)))
    with pytest.raises(TypeError):
        rb8.add(frames[0].astype(np.float32), act[0], 0.0, nxt[0], False)


def test_stream_hint_is_per_stream_and_thread_safe(nat, tmp_path):
    """The dispatch hint lives with the stream (aid_stream_hint), not the process: two host threads launch the same GEMM concurrently on
    two streams; only the hinted stream's launches take the 35 KB-LDS variant (profiling label 15), the other's never do, and both get
    the same numbers.  (The reference runs collector inference from a daemon thread on its own stream: utils/async_collector.py:388,451.)"""
    import csv
    import threading
    rng = np.random.default_rng(0)
    M, N, K = 256, 256, 512      # (N >= 512 at 129..1024 rows goes to the one-workgroup-per-CU kernel whatever the hint)
    A, W = dev(rng.standard_normal((M, K))), dev(rng.standard_normal((N, K)) / 16)
    s_hint, s_plain = torch.cuda.Stream(), torch.cuda.Stream()
    nat.stream_hint(s_hint, 1)
    outs = {}
    nat.prof_enable(True); nat.prof_reset()

    def run(name, stream, reps):
        with torch.cuda.stream(stream):
            o = torch.zeros(M, N, device="cuda")
            for _ in range(reps):
                nat.gemm(A, W, o, M, N, K, K, K, N)
            outs[name] = o
    try:
        th = [threading.Thread(target=run, args=("hint", s_hint, 40)), threading.Thread(target=run, args=("plain", s_plain, 24))]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        path = str(tmp_path / "launches.csv")
        nat.prof_dump(path)
    finally:
        nat.prof_enable(False); nat.prof_reset()
        nat.stream_hint(s_hint, 0)
    variants = [int(r["variant"]) for r in csv.DictReader(open(path)) if int(r["M"]) == M and int(r["N"]) == N]
    assert variants.count(15) == 40 and len(variants) == 64 and 15 not in [v for v in variants if v != 15]
    assert torch.equal(outs["hint"], outs["plain"]) or torch.allclose(outs["hint"], outs["plain"], rtol=1e-5, atol=1e-5)
    # cleared: the formerly hinted stream dispatches like any other
    nat.prof_enable(True); nat.prof_reset()
    try:
        run("after", s_hint, 3); torch.cuda.synchronize(); nat.prof_dump(path)
    finally:
        nat.prof_enable(False); nat.prof_reset()
    assert all(int(r["variant"]) != 15 for r in csv.DictReader(open(path)))


@pytest.mark.parametrize("M,N,K,act,skip", [(2048, 1024, 1024, "silu", True), (4100, 512, 1024, "silu", False), (2560, 512, 384, "relu", False),
                                             (2049, 1024, 72, "none", True), (25600, 1024, 1024, "silu", True)])
def test_linear_ln_act_full_row_kernel(nat, M, N, K, act, skip):
    """Linear -> LayerNorm -> activation (+ skip) with the LayerNorm in the GEMM's epilogue (csrc/gemm_rowln.hip) against torch fp32 on
    the same inputs (ragged row counts, K not a multiple of the stage depth, non-trivial affine, a row-dependent offset so that the
    mean matters).  Reference op chain: the decoder stages of core/active_inference.py:237-242."""
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (0.5 * rng.standard_normal(N) + 0.3).astype(np.float32)
    ga = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32); be = (0.1 * rng.standard_normal(N)).astype(np.float32)
    x[:, 0] += np.linspace(-3, 3, M).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32) if skip else None
    T = torch.from_numpy
    lin = torch.nn.functional.linear(T(x).double(), T(W).double(), T(b).double())
    ref = torch.nn.functional.layer_norm(lin, (N,), T(ga).double(), T(be).double(), 1e-5)
    ref = {"silu": torch.nn.functional.silu, "relu": torch.relu, "none": lambda v: v}[act](ref)
    if skip:
        ref = ref + T(add).double()
    got = nat.linear_ln_act(dev(x), dev(W), dev(b), dev(ga), dev(be), act, dev(add) if skip else None)
    np.testing.assert_allclose(got.cpu().numpy(), ref.float().numpy(), rtol=2e-5, atol=2e-5)
    if M <= 4100:      # dropout mask between the activation and the skip (train-mode Linear-LN-act-Dropout stages)
        mask = ((rng.random((M, N)) >= 0.1) / 0.9).astype(np.float32)
        gm = nat.linear_ln_act(dev(x), dev(W), dev(b), dev(ga), dev(be), act, dev(add) if skip else None, dev(mask))
        refm = (ref - (T(add).double() if skip else 0)) * T(mask).double() + (T(add).double() if skip else 0)
        np.testing.assert_allclose(gm.cpu().numpy(), refm.float().numpy(), rtol=2e-5, atol=2e-5)
    # and against the two-kernel path the small-M layers keep using (GEMM, then the row-wise LayerNorm kernel)
    two = torch.empty(M, N, device="cuda")
    nat.gemm(dev(x), dev(W), two, M, N, K, K, K, N, bias=dev(b))
    ref32 = torch.nn.functional.layer_norm(two, (N,), dev(ga), dev(be), 1e-5)
    ref32 = {"silu": torch.nn.functional.silu, "relu": torch.relu, "none": lambda v: v}[act](ref32) + (dev(add) if skip else 0)
    np.testing.assert_allclose(got.cpu().numpy(), ref32.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M", [129, 256, 300, 512, 768, 1000, 1024, 1500, 3072, 4100, 8192])
def test_gemm_mid_and_adaln_sum(nat, M):
    """The mid-M pipeline of the reverse loop (csrc/gemm_mid.hip): one-workgroup-per-CU GEMM with GELU / bias / residual epilogue, its split-K
    slab form, and the adaLN that folds the slabs into the residual stream, against float64 torch.  Row counts: every BM the kernel
    instantiates (32 / 64 / 96 / 128-row tiles), ragged ones included, and beyond 1024 rows (128-row tiles, as many as the rows need: the 3 072-row
    merged loops of a batch-1024 step).  Reference ops: models/score_networks.py:214-235 (DiTBlock.forward)."""
    rng = np.random.default_rng(M)
    D = 512
    T = lambda a: torch.from_numpy(a).double()
    x = rng.standard_normal((M, D)).astype(np.float32)
    W1 = (rng.standard_normal((4 * D, D)) / np.sqrt(D)).astype(np.float32); b1 = rng.standard_normal(4 * D).astype(np.float32)
    W2 = (rng.standard_normal((D, 4 * D)) / np.sqrt(4 * D)).astype(np.float32); b2 = rng.standard_normal(D).astype(np.float32)
    res = rng.standard_normal((M, D)).astype(np.float32)
    # ksplit = 1: GELU(x W1^T + b1)
    u = nat.gemm_mid(dev(x), dev(W1), dev(b1), "gelu")
    u_ref = torch.nn.functional.gelu(T(x) @ T(W1).T + T(b1))
    np.testing.assert_allclose(u.cpu().numpy(), u_ref.float().numpy(), rtol=1e-5, atol=4e-6)      # K = 512 fp32 products, up to 16.8 M of them
    # ksplit = 1 with residual, no activation, N not a multiple of the 64-column tile
    Wn = W2[:200, :D].copy(); bn = b2[:200].copy(); rn = res[:, :200].copy()
    y = nat.gemm_mid(dev(x), dev(Wn), dev(bn), "none", dev(rn))
    np.testing.assert_allclose(y.cpu().numpy(), (T(x) @ T(Wn).T + T(bn) + T(rn)).float().numpy(), rtol=1e-5, atol=2e-6)
    # split-K slabs: the ordered sum of the 4 slices is the product
    un = u_ref.float().numpy()
    slab = nat.gemm_mid(dev(un), dev(W2), ksplit=4)
    assert slab.shape == (4, M, D)
    prod = T(un) @ T(W2).T
    np.testing.assert_allclose(slab.double().sum(0).cpu().numpy(), prod.numpy(), rtol=1e-5, atol=5e-6)      # K = 2048 fp32 products
    for yk in range(4):     # each slab is exactly its K slice
        sl = slice(yk * D, (yk + 1) * D)
        np.testing.assert_allclose(slab[yk].cpu().numpy(), (T(un[:, sl]) @ T(W2[:, sl]).T).float().numpy(), rtol=1e-5, atol=5e-6)
    # adaLN over residual + bias + slabs
    mod = (0.3 * rng.standard_normal((M, 3 * 2 * D))).astype(np.float32)         # a row of several modulations: take the middle one
    mview = dev(mod)[:, 2 * D:4 * D]
    xo, yo = nat.adaln_sum(dev(res), slab, dev(b2), mview)
    x_ref = T(res) + T(b2) + prod
    sc, sh = T(mod[:, 2 * D:3 * D]), T(mod[:, 3 * D:4 * D])
    y_ref = torch.nn.functional.layer_norm(x_ref, (D,), None, None, 1e-5) * (1 + sc) + sh
    np.testing.assert_allclose(xo.cpu().numpy(), x_ref.float().numpy(), rtol=1e-5, atol=6e-6)
    np.testing.assert_allclose(yo.cpu().numpy(), y_ref.float().numpy(), rtol=2e-5, atol=2e-5)
    # no slabs, no bias, no x output: plain adaLN
    _, y0 = nat.adaln_sum(dev(res), None, None, mview, want_x=False)
    y0_ref = torch.nn.functional.layer_norm(T(res), (D,), None, None, 1e-5) * (1 + sc) + sh
    np.testing.assert_allclose(y0.cpu().numpy(), y0_ref.float().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M", [1, 7, 32, 33, 64, 100, 128])
def test_small_rows_split_k_pipeline(nat, M):
    """The <= 128-row reverse loop's products (act() / collector batches, 32-row shards of a strong-scaling run; csrc/gemm_mid.hip, gemm_w4_kernel): 32 x 16
    tiles, the waves of a workgroup a part of the K slice each, operands straight into registers - x Wov^T in 2 raw slices, x W1^T + GELU complete,
    u W2^T in 4 (or 8) raw slices folded by the adaLN - against float64 torch.  Reference ops: models/score_networks.py:214-235."""
    rng = np.random.default_rng(1000 + M)
    D = 512
    T = lambda a: torch.from_numpy(a).double()
    x = rng.standard_normal((M, D)).astype(np.float32)
    W1 = (rng.standard_normal((4 * D, D)) / np.sqrt(D)).astype(np.float32); b1 = rng.standard_normal(4 * D).astype(np.float32)
    W2 = (rng.standard_normal((D, 4 * D)) / np.sqrt(4 * D)).astype(np.float32); b2 = rng.standard_normal(D).astype(np.float32)
    res = rng.standard_normal((M, D)).astype(np.float32)
    pre = T(x) @ T(W1).T
    u_ref = torch.nn.functional.gelu(pre + T(b1))
    prod = u_ref @ T(W2).T
    un = u_ref.float().numpy()
    s2 = nat.gemm_w4(dev(un), dev(W2), ksplit=4)                   # u W2^T in 4 raw slices (8 waves x 64 per slice)
    mod = (0.3 * rng.standard_normal((M, 2 * D))).astype(np.float32)
    xo, yo = nat.adaln_sum(dev(res), s2, dev(b2), dev(mod))                         # the adaLN that folds them
    x_ref = T(res) + T(b2) + prod
    y_ref = torch.nn.functional.layer_norm(x_ref, (D,), None, None, 1e-5) * (1 + T(mod[:, :D])) + T(mod[:, D:])
    np.testing.assert_allclose(xo.cpu().numpy(), x_ref.float().numpy(), rtol=1e-5, atol=8e-6)
    np.testing.assert_allclose(yo.cpu().numpy(), y_ref.float().numpy(), rtol=2e-5, atol=3e-5)
    s16 = nat.gemm_mid(dev(un), dev(W2), ksplit=16)                                  # gemm_mid's 32-row tiles accept any row count too (16 x 128-deep slices)
    xo16, _ = nat.adaln_sum(dev(res), s16, dev(b2), dev(mod))
    np.testing.assert_allclose(xo16.cpu().numpy(), x_ref.float().numpy(), rtol=1e-5, atol=8e-6)
    s3 = nat.gemm_mid(dev(x), dev(W2[:, :D].copy()), ksplit=4)                      # the x Wov^T shape: 512 x 512 in 4 slices
    np.testing.assert_allclose(s3.double().sum(0).cpu().numpy(), (T(x) @ T(W2[:, :D]).T).numpy(), rtol=1e-5, atol=4e-6)
    # gemm_w4 (what the loop runs): 32 x 16 tiles, four waves a quarter of the K slice each
    u4 = nat.gemm_w4(dev(x), dev(W1), dev(b1), "gelu")                              # complete, bias + GELU epilogue (K = 512: 8 waves x 64)
    np.testing.assert_allclose(u4.cpu().numpy(), u_ref.float().numpy(), rtol=1e-5, atol=4e-6)
    for ks in (4, 8):                                                               # u W2^T in 4 (128 per wave) / 8 (64 per wave) raw slices
        s4 = nat.gemm_w4(dev(un), dev(W2), ksplit=ks)
        assert s4.shape == (ks, M, D)
        np.testing.assert_allclose(s4.double().sum(0).cpu().numpy(), (T(un) @ T(W2).T).numpy(), rtol=1e-5, atol=6e-6)
        sl = slice((ks - 1) * (4 * D // ks), 4 * D)
        np.testing.assert_allclose(s4[ks - 1].cpu().numpy(), (T(un[:, sl]) @ T(W2[:, sl]).T).float().numpy(), rtol=1e-5, atol=4e-6)
    s5 = nat.gemm_w4(dev(x), dev(W2[:, :D].copy()), ksplit=2)                        # x Wov^T in 2 slices (64 per wave)
    np.testing.assert_allclose(s5.double().sum(0).cpu().numpy(), (T(x) @ T(W2[:, :D]).T).numpy(), rtol=1e-5, atol=4e-6)
    h4 = nat.gemm_w4(dev(x), dev(W1[:200].copy()), dev(b1[:200].copy()), "silu")     # N not a multiple of 16 (ragged last tile), SiLU
    np.testing.assert_allclose(h4.cpu().numpy(), torch.nn.functional.silu(T(x) @ T(W1[:200]).T + T(b1[:200])).float().numpy(), rtol=1e-5, atol=4e-6)


@pytest.mark.parametrize("M,N,K,act", [(4096, 512, 384, "relu"), (4100, 1024, 72, "none"), (5000, 1536, 512, "gelu"), (25600, 512, 512, "silu")])
def test_tall_linear_full_row_blocks(nat, M, N, K, act):
    """aid_gemm_f32 routes tall Linear forwards whose width is a multiple of 512 to the 64-row x 512-column full-K workgroups of
    csrc/gemm_rowln.hip (plain epilogue: bias, activation, dropout mask, residual): ragged rows, a K tail, several column blocks."""
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).double()
    pre = T(x) @ T(W).T + T(b)
    ref = {"relu": torch.relu, "silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu, "none": lambda v: v}[act](pre) + T(R)
    out = torch.zeros(M, N, device="cuda")
    import csv, tempfile
    nat.prof_enable(True); nat.prof_reset()
    try:
        nat.gemm(dev(x), dev(W), out, M, N, K, K, K, N, bias=dev(b), act=act, residual=dev(R), ldr=N)
        torch.cuda.synchronize()
        path = tempfile.mktemp(suffix=".csv"); nat.prof_dump(path)
    finally:
        nat.prof_enable(False); nat.prof_reset()
    x6 = os.environ.get("AID_GEMM_X6") == "1" and K % 32 == 0                          # suite run with the six-term bf16 form switched on
    assert [int(r["variant"]) for r in csv.DictReader(open(path))] == [29 if x6 else 23]          # the full-row kernel ran, not the 64x64 tile
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,K", [(4100, 1024), (25600, 512), (2048, 80), (4096, 72)])
def test_full_row_kernels_with_16_deep_stages_under_the_stream_hint(nat, M, K):
    """aid_stream_hint bit 1: the 512-wide full-row kernels launched on that stream take 16-deep stages (74 KB of LDS, so that another stream's
    workgroups can share the CU).  Same results as on an un-hinted stream - LayerNorm form with mask and skip, and the plain form through
    aid_gemm_f32 - and the profile records which instantiation ran (layout bit 5); K = 72 is not a multiple of 16 and keeps the 32-deep kernel
    with its K tail."""
    import csv, tempfile
    if os.environ.get("AID_GEMM_X6") == "1" and K % 32 == 0:
        pytest.skip("suite run with the six-term bf16 form switched on: it replaces the float32 kernels this test tells apart")
    rng = np.random.default_rng(M + K)
    N = 512
    x = dev(rng.standard_normal((M, K))); W = dev(rng.standard_normal((N, K)) / np.sqrt(K)); b = dev(rng.standard_normal(N))
    ga = dev(1 + 0.2 * rng.standard_normal(N)); be = dev(0.1 * rng.standard_normal(N)); add = dev(rng.standard_normal((M, N)))
    mask = dev((rng.random((M, N)) >= 0.1) / 0.9)
    want_ln = nat.linear_ln_act(x, W, b, ga, be, "silu", add, mask)
    want_plain = torch.zeros(M, N, device="cuda"); plain_ok = M >= 4096
    if plain_ok:
        nat.gemm(x, W, want_plain, M, N, K, K, K, N, bias=b, act="relu", residual=add, ldr=N)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    nat.stream_hint(s, 2)
    nat.prof_enable(True); nat.prof_reset()
    try:
        with torch.cuda.stream(s):
            got_ln = nat.linear_ln_act(x, W, b, ga, be, "silu", add, mask)
            got_plain = torch.zeros(M, N, device="cuda")
            if plain_ok:
                nat.gemm(x, W, got_plain, M, N, K, K, K, N, bias=b, act="relu", residual=add, ldr=N)
        torch.cuda.synchronize()
        path = tempfile.mktemp(suffix=".csv"); nat.prof_dump(path)
    finally:
        nat.prof_enable(False); nat.prof_reset(); nat.stream_hint(s, 0)
    rows = [(int(r["variant"]), int(r["layout"])) for r in csv.DictReader(open(path)) if int(r["variant"]) in (21, 23)]
    deep16 = K % 16 == 0
    assert rows[0][0] == 21 and bool(rows[0][1] & 32) == deep16, rows
    if plain_ok:
        assert rows[1][0] == 23 and bool(rows[1][1] & 32) == deep16, rows
    # a different stage depth is a different summation order along K: equal to fp32 rounding, not bit for bit
    np.testing.assert_allclose(got_ln.cpu().numpy(), want_ln.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got_plain.cpu().numpy(), want_plain.cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_stream_hint_slots_are_freed_when_cleared(nat):
    """The library's per-stream hint table has 64 slots; clearing a stream's flags frees its slot, so an agent that is driven from many
    different streams over its lifetime (each set, then cleared when it moves on) never runs out."""
    streams = [0x7000000 + 64 * i for i in range(80)]           # more distinct handles than the table has slots (the table never dereferences them)
    for s in streams:
        nat.stream_hint(s, 2); nat.stream_hint(s, 0)           # without the free, the 65th set would be refused
    for s in streams[:16]:                                       # and live hints still fit afterwards (other tests' side streams hold some slots)
        nat.stream_hint(s, 1)
    for s in streams[:16]:
        nat.stream_hint(s, 0)


def test_grouped_weight_gradients(nat):
    """csrc/gemm_group.hip: every dW += dY^T X (and db += colsum dY) of a backward pass in one launch.  Ragged problems (N or K = 17, 6, 1: the
    score head / latent projection / cte0 shapes), a contraction that is not a multiple of the 32-row stage, two segments into one gradient
    (the two adjoint streams of the gradient-penalty pass), more problems than one launch holds (the queue flushes in between), accumulation
    on top of existing gradients.  Reference: fp64 matmul."""
    rng = np.random.default_rng(5)
    shapes = [(256, 512, 512), (256, 2048, 512), (256, 512, 2048), (256, 17, 256), (256, 512, 17), (250, 128, 1), (37, 6, 256), (256, 13312, 512),
              (512, 256, 256), (12800, 512, 512), (6001, 6, 256), (5000, 96, 17)] + [(64, 96, 40)] * 40      # the three long ones are split into row chunks
    dW, dy, x, db, ref_w, ref_b = [], [], [], [], [], []
    for (m, n, k) in shapes:
        g, a = rng.standard_normal((m, n)).astype(np.float32), rng.standard_normal((m, k)).astype(np.float32)
        w0, b0 = rng.standard_normal((n, k)).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        dW.append(torch.from_numpy(w0.copy()).cuda()); dy.append(torch.from_numpy(g).cuda()); x.append(torch.from_numpy(a).cuda())
        db.append(torch.from_numpy(b0.copy()).cuda())
        ref_w.append(w0.astype(np.float64) + g.astype(np.float64).T @ a.astype(np.float64)); ref_b.append(b0.astype(np.float64) + g.astype(np.float64).sum(0))
    # a second segment for problem 1 (same dW object, listed right behind it; no bias contribution)
    g2, a2 = rng.standard_normal((256, 2048)).astype(np.float32), rng.standard_normal((256, 512)).astype(np.float32)
    dW.insert(2, dW[1]); dy.insert(2, torch.from_numpy(g2).cuda()); x.insert(2, torch.from_numpy(a2).cuda()); db.insert(2, None)
    ref_w[1] = ref_w[1] + g2.astype(np.float64).T @ a2.astype(np.float64)
    nat.gemm_group_tn(dW, dy, x, [d if d is not None else None for d in db], scratch=torch.empty(8 * (512 * 512 + 512) + 8 * (6 * 256 + 6) + 100000, device="cuda"))
    torch.cuda.synchronize()
    outs_w = [w for i, w in enumerate(dW) if i != 2]; outs_b = [b for i, b in enumerate(db) if i != 2]
    for (m, n, k), w, b, rw, rb in zip(shapes, outs_w, outs_b, ref_w, ref_b):
        tol = 2e-5 * np.sqrt(m) * (2 if (n, k) == (2048, 512) else 1)      # fp32 accumulation over m products of O(1) terms
        np.testing.assert_allclose(w.cpu().numpy(), rw, rtol=1e-5, atol=tol, err_msg=str((m, n, k)))
        np.testing.assert_allclose(b.cpu().numpy(), rb, rtol=1e-5, atol=tol, err_msg=str((m, n, k)))


@pytest.mark.parametrize("M,N,K,lay,act", [(2560, 512, 512, "nt", "none"), (2560, 512, 512, "nn", "none"), (2560, 512, 512, "nt", "relu"), (256, 2048, 512, "nt", "gelu"),
                                           (256, 2048, 512, "nn", "none"), (512, 2048, 512, "nn", "none"), (1280, 1024, 256, "nt", "silu"), (3200, 512, 128, "nn", "none")])
def test_gemm_mid_routed_shapes(nat, M, N, K, lay, act):
    """Shapes gemm_f32 hands to the one-workgroup-per-CU kernel (csrc/gemm_mid.hip): 80 x 64 tiles for the rollout's 2 560-row layers, 8 row
    tiles for 129..1024 rows x >= 2048 columns; W as [N][K] (forward) and as [K][N] (data gradient), bias + activation + residual."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N) if lay == "nn" else (N, K)).astype(np.float32) / np.sqrt(K)
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ (W if lay == "nn" else W.T).astype(np.float64) + b
    f = {"none": lambda x: x, "relu": lambda x: np.maximum(x, 0), "silu": lambda x: x / (1 + np.exp(-x)),
         "gelu": lambda x: 0.5 * x * (1 + np.vectorize(__import__("math").erf)(x / np.sqrt(2)))}[act]
    ref = f(ref) + R
    C = torch.empty(M, N, device="cuda")
    nat.gemm(dev(A), dev(W), C, M, N, K, K, W.shape[1], N, 0, 1 if lay == "nn" else 0, bias=dev(b), act=act, residual=dev(R), ldr=N)
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(2560, 512, 17), (300, 70, 6), (64, 256, 1), (2560, 256, 2)])
def test_gemm_small_k_data_gradient_layout(nat, tmp_path, M, N, K):
    """K <= 32 with W stored [K][N] (the data gradient through a layer with <= 32 outputs, e.g. dh = dmean W_mean2 in the policy BPTT): the
    VALU small-K kernel (variant 3) stages that layout too instead of leaving the shape to the unaligned 64 x 64 MFMA tile kernel."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((K, N)).astype(np.float32); C0 = rng.standard_normal((M, N)).astype(np.float32)
    C = dev(C0.copy())
    nat.prof_enable(True); nat.prof_reset()
    nat.gemm(dev(A), dev(W), C, M, N, K, K, N, N, 0, 1, accumulate=1)
    torch.cuda.synchronize(); nat.prof_dump(str(tmp_path / "prof.csv")); nat.prof_enable(False); nat.prof_reset()
    import csv
    recs = list(csv.DictReader(open(tmp_path / "prof.csv")))
    assert len(recs) == 1 and int(recs[0]["variant"]) == 3, recs
    np.testing.assert_allclose(C.cpu().numpy(), A.astype(np.float64) @ W.astype(np.float64) + C0, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,N,K,lay,act,acc", [(2560, 17, 512, "nn", "none", 0), (2560, 6, 256, "nt", "relu", 0), (2560, 1, 512, "nt", "none", 1), (257, 23, 512, "nn", "none", 0),
                                               (100, 16, 1024, "nt", "silu", 0), (96, 32, 512, "nn", "none", 0), (4099, 2, 256, "nt", "relu", 0), (64, 17, 384, "nn", "none", 1), (300, 9, 68, "nt", "none", 0)])
def test_gemm_thin_outputs(nat, tmp_path, M, N, K, lay, act, acc):
    """N <= 32 output columns at 33..8192 rows (csrc/gemm_thin.hip: W transposed into LDS, a wave per four rows, fold8 reductions): the small
    networks' heads over the rollout's 2 560 rows and the data gradients into their 17- / 23-wide inputs (W as stored, [K][N] with ld = N).
    Ragged M, K not a multiple of 256, bias + activation + residual + accumulate; the profile names the route (variant 25)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N) if lay == "nn" else (N, K)).astype(np.float32) / np.sqrt(K)
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32); C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ (W if lay == "nn" else W.T).astype(np.float64) + b
    f = {"none": lambda x: x, "relu": lambda x: np.maximum(x, 0), "silu": lambda x: x / (1 + np.exp(-x)), "tanh": np.tanh}[act]
    ref = f(ref) + R + (C0 if acc else 0)
    C = dev(C0.copy())
    nat.prof_enable(True); nat.prof_reset()
    nat.gemm(dev(A), dev(W), C, M, N, K, K, W.shape[1], N, 0, 1 if lay == "nn" else 0, bias=dev(b), act=act, residual=dev(R), ldr=N, accumulate=acc)
    torch.cuda.synchronize(); nat.prof_dump(str(tmp_path / "prof.csv")); nat.prof_enable(False); nat.prof_reset()
    import csv
    recs = list(csv.DictReader(open(tmp_path / "prof.csv")))
    assert len(recs) == 1 and int(recs[0]["variant"]) == 25, recs
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K,lay,act,acc", [(256, 512, 2048, "nt", "none", 0), (256, 512, 2048, "nn", "none", 1), (512, 512, 2048, "nn", "none", 0),
                                               (256, 1024, 1024, "nt", "gelu", 0), (200, 512, 13312, "nn", "none", 0), (768, 256, 1024, "nt", "silu", 0)])
def test_gemm_split_k_folded_in_kernel(nat, tmp_path, M, N, K, lay, act, acc):
    """gemm_f32's split-K route for 129..1024 rows x <= 1024 columns, K >= 1024 (csrc/gemm_mid.hip with tile counters; what aid_elbo_step's passes
    take for the DiT blocks' 512-wide products): K slices over ~256 workgroups, the last slice to arrive folds the others in slice order and
    applies bias / activation / residual / accumulate.  Checked against fp64, for bit-identical repeats (the fold order is fixed whoever folds),
    for the profile naming the route (variant 24) and for the tile counters being back at zero (the same scratch serves every call)."""
    rng = np.random.default_rng(M + N + K)
    f = {"none": lambda x: x, "silu": lambda x: x / (1 + np.exp(-x)), "gelu": lambda x: 0.5 * x * (1 + np.vectorize(__import__("math").erf)(x / np.sqrt(2)))}[act]
    scratch = nat.gemm_scratch()
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32); C0 = rng.standard_normal((M, N)).astype(np.float32)
    bd, Rd = dev(b), dev(R)

    def launch(Ad, Wd, scr=scratch):
        C = dev(C0.copy())
        nat.gemm(Ad, Wd, C, M, N, K, K, Wd.shape[1], N, 0, 1 if lay == "nn" else 0, bias=bd, act=act, residual=Rd, ldr=N, accumulate=acc, scratch=scr)
        return C
    # A test that can FAIL on a stale or early slice read (the fold's hand-off across XCDs is ordered by sc1 accesses + vmcnt, not by the C++
    # memory model): every launch gets NEW operands, so the slices a previous launch left in the scratch hold wrong values, and the whole
    # scratch - slices and counters; the counters are zeroed again when the call opens its scope - is poisoned with NaN in between.
    nat.prof_enable(True); nat.prof_reset()
    outs, refs, ops = [], [], []
    for i in range(3):
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = rng.standard_normal((K, N) if lay == "nn" else (N, K)).astype(np.float32) / np.sqrt(K)
        ref = A.astype(np.float64) @ (W if lay == "nn" else W.T).astype(np.float64) + b
        refs.append(f(ref) + R + (C0 if acc else 0))
        ops.append((dev(A), dev(W)))
        scratch.fill_(float("nan"))
        outs.append(launch(*ops[-1]))
    torch.cuda.synchronize()
    nat.prof_dump(str(tmp_path / "prof.csv"))
    nat.prof_enable(False); nat.prof_reset()
    import csv
    recs = list(csv.DictReader(open(tmp_path / "prof.csv")))
    assert len(recs) == 3 and all(int(r["variant"]) == 24 for r in recs), recs
    for o, ref in zip(outs, refs):
        o = o.cpu().numpy()
        assert np.isfinite(o).all()
        np.testing.assert_allclose(o, ref, rtol=2e-5, atol=2e-5 * np.sqrt(K))
    # bit-identical repeats of the first problem, each on a freshly poisoned scratch and after other problems have used it (the fold order is
    # fixed whoever folds); the tile counters are back at zero after every launch (the same scratch serves every call)
    for _ in range(2):
        scratch.fill_(float("nan"))
        assert torch.equal(launch(*ops[0]), outs[0])
    # without scratch the dispatcher takes another kernel: same value to rounding
    C = dev(C0.copy())
    nat.gemm(ops[0][0], ops[0][1], C, M, N, K, K, ops[0][1].shape[1], N, 0, 1 if lay == "nn" else 0, bias=bd, act=act, residual=Rd, ldr=N, accumulate=acc)
    np.testing.assert_allclose(C.cpu().numpy(), outs[0].cpu().numpy(), rtol=2e-5, atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize("Cin,Cout,B,Hi,stride", [(9, 32, 4, 84, 2), (32, 64, 4, 42, 1), (64, 128, 2, 42, 1), (128, 256, 2, 20, 1), (5, 70, 3, 14, 1), (3, 3, 2, 12, 2),
                                                   (32, 32, 3, 21, 1), (64, 32, 2, 30, 1), (32, 3, 2, 17, 1), (16, 80, 3, 13, 2),
                                                   (32, 128, 40, 42, 1), (16, 160, 38, 42, 1),         # >= 65 536 pixels x >= 128 channels: the 128 x 128 tile
                                                   (128, 128, 3, 21, 1), (192, 256, 1, 16, 1)])        # weight gradient on 128 x 128 tap-major tiles: odd stage counts, a tile across two taps
def test_conv3_implicit_gemm(nat, Cin, Cout, B, Hi, stride):
    """csrc/conv_igemm.hip against torch's conv2d on the CPU (fp64): forward (border taps, stride 2, ragged channel counts), the data
    gradient through the flipped weights, the weight gradient (slab-split contraction over B Ho Wo) - and, where the channel counts allow it, the
    TAP-MAJOR kernels the encoder / ConvDecoder run (weights re-ordered to k' = tap Cin + ci; forward and data gradient), every tile shape."""
    import torch.nn.functional as F
    rng = np.random.default_rng(Cin * 100 + Cout)
    x = rng.standard_normal((B, Cin, Hi, Hi)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    K = 9 * Cin; ldw = (K + 3) // 4 * 4
    W = np.zeros((Cout, ldw), np.float32); W[:, :K] = w.reshape(Cout, K)
    xt = torch.from_numpy(x).double().requires_grad_(True); wt = torch.from_numpy(w).double().requires_grad_(True)
    ref = F.conv2d(xt, wt, stride=stride, padding=1)
    x_cnhw = dev(np.ascontiguousarray(x.transpose(1, 0, 2, 3)))
    out = nat.conv3_igemm(dev(W), x_cnhw, stride)
    np.testing.assert_allclose(out.cpu().numpy().transpose(1, 0, 2, 3), ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    dy = rng.standard_normal(tuple(ref.shape)).astype(np.float32)
    ref.backward(torch.from_numpy(dy).double())
    dy_cnhw = dev(np.ascontiguousarray(dy.transpose(1, 0, 2, 3)))
    if (B * ref.shape[2] * ref.shape[3]) % 4 == 0:
        for mode in (15, 16, 14):      # 128 x 128 weight-gradient tiles where the product takes them / wherever the kernel can run (Cin % 64 == 0, Cout % 128 == 0) / never
            nat.lib().aid_conv_debug(mode)
            dW = nat.conv3_igemm_dw(dy_cnhw, x_cnhw, ldw, stride, slabs=5)
            np.testing.assert_allclose(dW.cpu().numpy()[:, :K].reshape(w.shape), wt.grad.numpy(), rtol=2e-5, atol=2e-5 * np.sqrt(B * ref.shape[2] * ref.shape[3]))
            assert float(dW[:, K:].abs().max()) == 0.0 if ldw > K else True
        nat.lib().aid_conv_debug(15)
    if stride == 1:
        Wt = nat.conv3_flip_weights(dev(W), Cin)
        dx = nat.conv3_igemm(Wt, dy_cnhw, 1)
        np.testing.assert_allclose(dx.cpu().numpy().transpose(1, 0, 2, 3), xt.grad.numpy(), rtol=2e-5, atol=5e-5)
    tm_ok = lambda ci, co: ci % (32 if co <= 32 else 16) == 0
    if tm_ok(Cin, Cout):
        out_tm = nat.conv3_igemm_tm(dev(W), x_cnhw, stride)
        np.testing.assert_allclose(out_tm.cpu().numpy().transpose(1, 0, 2, 3), ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    if stride == 1 and tm_ok(Cout, Cin):
        dx_tm = nat.conv3_igemm_tm(dev(W), dy_cnhw, 1, flip_transpose=True)
        np.testing.assert_allclose(dx_tm.cpu().numpy().transpose(1, 0, 2, 3), xt.grad.numpy(), rtol=2e-5, atol=5e-5)


@pytest.mark.parametrize("B", [1, 3, 8, 16])
def test_small_batch_loop_with_adaln_folded_into_its_gemms(nat, B):
    """<= 16 rows (act(), the collector's smallest batches): the reverse loop with the adaLN launches folded into the GEMMs that consume them
    (csrc/gemm_mid.hip, gemm_w4ln_kernel: 20 launches per denoise step instead of 32) against the same loop with separate adaLN kernels - the SAME arithmetic
    in the same order, so bit-identical latents and trajectory - and, through that loop, against everything the unfused loop is tested against (loop goldens,
    oracle).  Reference: models/score_networks.py:214-235, core/diffusion.py:176-206."""
    from oracle import aid_oracle as orc
    from oracle.weights import draw_params, param_specs
    O, D, N = 17, 512, 6
    p = draw_params(param_specs(O, 6, O, D, N), 77, orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    tab = nat.score_net_table(pd, "latent_score_network.", O, O, D)
    coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
    rng = np.random.default_rng(B)
    obs, z0, eps = dev(rng.standard_normal((B, O))), dev(rng.standard_normal((B, O))), dev(rng.standard_normal((N, B, O)))
    outs = []
    for mode in (-10, -11):
        nat.lib().aid_denoise_debug(mode)
        z, traj = nat.denoise_loop(tab, coefs, obs, N, z0, eps, None, train=False, keep_traj=True)
        torch.cuda.synchronize()
        outs.append((z.clone(), traj.clone()))
    nat.lib().aid_denoise_debug(-10)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = orc.generate_latent_trajectory({k: v for k, v in p.items()}, orc.schedule_tables(N), obs.cpu(), _FixedNoise(z0.cpu(), eps.cpu()), False)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-5)


class _FixedNoise:
    """The draws generate_latent_trajectory makes, served from given tensors: randn(B, L) first, then one randn per step t = N-1 .. 1."""
    def __init__(self, z0, eps):
        self.z0, self.eps, self.t = z0, eps, eps.shape[0] - 1
        self.first = True

    def randn(self, *shape):
        if self.first:
            self.first = False
            return self.z0.numpy()
        out = self.eps[self.t].numpy(); self.t -= 1
        return out


def test_main_stream_hint_changes_speed_not_bits(nat):
    """The hint an agent leaves on the calling stream (aid_stream_hint bit 1: 512-wide full-row kernels in 16-deep instead of 32-deep stages, so that a chain
    workgroup can share the CU) must not change results - otherwise a train_step's bits would depend on what ran earlier in the process on that stream
    (VERDICT r4 weak #3).  Both stage depths walk K in the same order inside one wave, so the tall Linear / Linear+LayerNorm+act products are bit-identical
    with and without it; bit 0 (the low-LDS small-GEMM variants) is only ever set on the library's own side / chain streams."""
    rng = np.random.default_rng(3)
    s = torch.cuda.Stream()
    outs = []
    for flags in (0, 2):
        nat.stream_hint(s, flags)
        with torch.cuda.stream(s):
            res = []
            for (M, N, K) in ((25600, 512, 512), (2560, 1024, 512), (12800, 512, 1024)):
                A, W, b = dev(rng.standard_normal((M, K)) if flags == 0 else saved[(M, N, K)][0]), None, None
                if flags == 0:
                    W, b = dev(rng.standard_normal((N, K)) / np.sqrt(K)), dev(rng.standard_normal(N))
                    saved[(M, N, K)] = (A.cpu().numpy(), W, b)
                else:
                    W, b = saved[(M, N, K)][1], saved[(M, N, K)][2]
                C_ = torch.empty(M, N, device="cuda")
                nat.gemm(A, W, C_, M, N, K, K, K, N, bias=b, act="silu")
                res.append(C_)
            torch.cuda.synchronize()
        outs.append(res)
    nat.stream_hint(s, 0)
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


saved = {}


@pytest.mark.parametrize("M,N,K,ln,skip", [(4096, 1024, 1024, True, True), (4160, 512, 1024, True, False), (4096, 512, 512, True, False), (4100, 1024, 512, False, False),
                                           (8192, 512, 384, False, False)])
def test_rowln_x6_is_at_least_as_accurate_as_the_float32_mfma(nat, M, N, K, ln, skip):
    """aid_gemm_emulation(1): the tall full-row GEMMs on the bf16 matrix pipe - each float32 operand split exactly into three bfloat16 values, the six leading
    partial products accumulated in float32 (csrc/gemm_rowln_x6.hip).  The claim that makes it admissible as float32 arithmetic: against a float64 evaluation of
    the same layer its error is not larger than that of the float32 MFMA kernel it replaces (max and rms), and it meets the same tolerance against the float32
    torch layer (rtol 1e-4 / atol 1e-6 for one forward layer ... the LayerNorm'd outputs here are O(1), so atol carries it: 2e-5)."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    ga = torch.rand(N, device="cuda") + 0.5; be = torch.randn(N, device="cuda")
    add = torch.randn(M, N, device="cuda") if skip else None
    pre = x.double() @ W.double().t() + b.double()
    if ln:
        exact = torch.nn.functional.silu(torch.nn.functional.layer_norm(pre, (N,), ga.double(), be.double(), 1e-5)) + (add.double() if skip else 0)
    else:
        exact = torch.relu(pre)
    out = {}
    mode0 = nat.gemm_emulation_mode()                    # restored afterwards: the suite may be running with AID_GEMM_X6=1
    try:
        for mode in (0, 1, 3):                           # 3: the first form (weights split in registers) for every shape - what a stream capture without a split buffer takes
            nat.gemm_emulation(mode)
            assert nat.gemm_emulation_mode() == (1 if mode else 0)
            if ln:
                y = nat.linear_ln_act(x, W, b, ga, be, "silu", add)
            else:
                y = torch.empty(M, N, device="cuda"); nat.gemm(x, W, y, M, N, K, K, K, N, bias=b, act="relu")
            e = (y.double() - exact).abs()
            out[mode] = (float(e.max()), float(e.pow(2).mean().sqrt()), y)
    finally:
        nat.gemm_emulation(mode0)
    (m0, r0, y0), (m1, r1, y1), (m3, r3, y3) = out[0], out[1], out[3]
    print(f"[{M} x {N} x {K} ln={ln}] vs float64: float32 MFMA max {m0:.3g} rms {r0:.3g} | six-term bf16 max {m1:.3g} rms {r1:.3g} | first form max {m3:.3g} rms {r3:.3g}")
    assert not torch.equal(y0, y1)                       # the flag does switch kernels
    for m_, r_ in ((m1, r1), (m3, r3)):
        assert r_ <= 1.05 * r0 and m_ <= 1.5 * m0, (m0, r0, m_, r_)
    np.testing.assert_allclose(y1.cpu().numpy(), exact.float().cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y3.cpu().numpy(), exact.float().cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_set_float32_matmul_emulation_switch(nat):
    """The public switch (aid_amd.set_float32_matmul_emulation) and what a belief-chain graph does with it: a graph captured under one arithmetic is not replayed
    under the other (the caches are keyed by the mode), and the latents of the two modes agree to float32 rounding."""
    import aid_amd
    mode0 = nat.gemm_emulation_mode()                    # restored afterwards: the suite may be running with AID_GEMM_X6=1
    with pytest.raises(ValueError):
        aid_amd.set_float32_matmul_emulation("tf32")
    try:
        aid_amd.set_float32_matmul_emulation("bf16x6"); assert nat.gemm_emulation_mode() == 1
        aid_amd.set_float32_matmul_emulation("off"); assert nat.gemm_emulation_mode() == 0
        O, A, D, N, B = 17, 6, 512, 50, 256
        cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=1, latent_dim=O, hidden_dim=D)
        cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)

        class _B:
            def __init__(self, shape): self.shape = shape

        class Env:
            observation_space, action_space = _B((O,)), _B((A,))
        ai = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=16)).active_inference
        from parity_noise import ParityNoise
        from oracle import aid_oracle as orc
        from oracle.weights import draw_params, param_specs
        p = draw_params(param_specs(O, A, O, D, N), 5, orc.schedule_tables(N))       # a fresh agent's score head is zero: its latents would not depend on the network
        sd0 = ai.state_dict()
        ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
        obs = [torch.randn(B, O, device="cuda"), torch.randn(B, O, device="cuda")]
        outs = {}
        for mode in ("off", "bf16x6", "off"):
            aid_amd.set_float32_matmul_emulation(mode)
            ai.noise = ParityNoise(77)                    # the same explicit draws for every run
            outs.setdefault(mode, []).append(ai.denoise_many(obs).clone())
        keys = {k[-1] for k in ai.__dict__["_chain_graphs"]}
        assert keys == {0, 1}, keys                      # one set of graphs per arithmetic
        assert torch.equal(outs["off"][0], outs["off"][1])      # switching back replays the float32 graphs: bit-identical
        a, b = outs["off"][0], outs["bf16x6"][0]
        assert not torch.equal(a, b)
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=1e-5)      # a 50-step loop
    finally:
        nat.gemm_emulation(mode0)
