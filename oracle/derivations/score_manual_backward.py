"""Derivation check (test infrastructure): the hand-written backward the HIP kernels implement,
written with plain tensor ops and NO autograd, compared against autograd of the oracle.

Gradient penalty  P = mean_b (||g_b|| - 1)^2,  g = d(sum s)/dz = J^T 1   (active_inference.py:709-729).
dP/dtheta = sum_b w_b . dg_b/dtheta with w = dP/dg held constant = d/dtheta [ sum (J w) ]: reverse mode
through a forward-mode (tangent) pass with tangent input z_dot = w.  The reverse pass carries two
adjoint streams: lam (adjoints of tangent activations) and mu (adjoints of primal activations, fed by
the second derivatives of SiLU/GELU/LayerNorm and by the adaLN scale products).
"""
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import aid_oracle as orc
from oracle.noise import NoiseStream
from oracle.weights import draw_params, param_specs

S = orc.SCORE


def silu_d(x):
    s = torch.sigmoid(x); return s * (1 + x * (1 - s))


def silu_dd(x):
    s = torch.sigmoid(x); ds = s * (1 - s)
    return ds * (2 + x * (1 - 2 * s))


def gelu_d(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def gelu_dd(x):
    return torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi) * (2 - x * x)


def ln_stats(x):
    m = x.mean(-1, keepdim=True); v = ((x - m) ** 2).mean(-1, keepdim=True)
    return m, 1.0 / torch.sqrt(v + 1e-5)


def ln_T(xh, r, d):
    """Jacobian of non-affine LN applied to d (symmetric operator)."""
    dc = d - d.mean(-1, keepdim=True)
    return r * (dc - xh * (xh * d).mean(-1, keepdim=True))


def ln_second(xh, r, hd, lam):
    """d/dh [ lam . T(h)[hd] ]  (derived in DESIGN.md)."""
    hc = hd - hd.mean(-1, keepdim=True); lc = lam - lam.mean(-1, keepdim=True)
    p = (xh * hd).mean(-1, keepdim=True); q = (xh * lam).mean(-1, keepdim=True)
    A = (lam * hc).mean(-1, keepdim=True)
    return -(r * r) * (xh * (A - 3 * p * q) + q * hc + p * lc)


def manual(p, z, t, obs, mask, d_score=None, z_dot=None, coef=1.0, nb=6):
    """Returns grads dict (name -> tensor) for: main path if d_score given (cotangent on score), or the
    tangent objective coef*sum(J z_dot) if z_dot given; plus dz (main) / score / tangent output."""
    D = p[S + "latent_proj.weight"].shape[0]
    G = {}
    def acc(n, v):
        G[n] = G.get(n, 0) + v
    # ---------------- forward (continuous branch) with saved activations
    fs = p[S + "time_embed.0.freq_scale"]
    half = 64; negc = -math.log(10000) / (half - 1)
    f0 = torch.exp(torch.arange(half, dtype=torch.float32) * negc)
    tv = t * 999.0
    arg = tv[:, None] * (f0 * fs)[None, :]
    sin_e = torch.cat([arg.sin(), arg.cos()], -1)
    te1_pre = F.linear(sin_e, p[S + "time_embed.1.weight"], p[S + "time_embed.1.bias"]); te1 = F.silu(te1_pre)
    t_sin = F.linear(te1, p[S + "time_embed.3.weight"], p[S + "time_embed.3.bias"])
    cin = (2 * t - 1)[:, None]
    c0_pre = F.linear(cin, p[S + "continuous_time_embed.0.weight"], p[S + "continuous_time_embed.0.bias"]); c0 = F.silu(c0_pre)
    c1_pre = F.linear(c0, p[S + "continuous_time_embed.2.weight"], p[S + "continuous_time_embed.2.bias"]); c1 = F.silu(c1_pre)
    c2 = F.linear(c1, p[S + "continuous_time_embed.4.weight"], p[S + "continuous_time_embed.4.bias"])
    ts = p[S + "time_scale"]
    temb = t_sin + ts * c2
    tw = torch.sqrt(1.0 / (1e-5 + t))[:, None]
    o0 = F.linear(obs, p[S + "obs_encoder.0.weight"], p[S + "obs_encoder.0.bias"]); m0, r0 = ln_stats(o0); xh0 = (o0 - m0) * r0
    y0 = xh0 * p[S + "obs_encoder.1.weight"] + p[S + "obs_encoder.1.bias"]; x1 = F.silu(y0) * mask
    o1 = F.linear(x1, p[S + "obs_encoder.4.weight"], p[S + "obs_encoder.4.bias"]); m1, r1 = ln_stats(o1); xh1 = (o1 - m1) * r1
    y1 = xh1 * p[S + "obs_encoder.5.weight"] + p[S + "obs_encoder.5.bias"]; x2 = F.silu(y1)
    o2 = F.linear(x2, p[S + "obs_encoder.7.weight"], p[S + "obs_encoder.7.bias"]); m2, r2 = ln_stats(o2); xh2 = (o2 - m2) * r2
    obs_emb = xh2 * p[S + "obs_encoder.8.weight"] + p[S + "obs_encoder.8.bias"]
    cond = temb + obs_emb; sc = F.silu(cond)
    def ada_names(i):
        if i == 2 * nb: return S + "norm_final.adaLN_modulation.1"
        return S + f"transformer_blocks.{i // 2}.norm{1 + i % 2}.adaLN_modulation.1"
    mods = [F.linear(sc, p[ada_names(i) + ".weight"], p[ada_names(i) + ".bias"]) for i in range(2 * nb + 1)]
    h = F.linear(z, p[S + "latent_proj.weight"], p[S + "latent_proj.bias"])
    saved = []
    for i in range(nb):
        b = S + f"transformer_blocks.{i}."
        Wv, bv = p[b + "attention.in_proj_weight"][2 * D:], p[b + "attention.in_proj_bias"][2 * D:]
        m_a, r_a = ln_stats(h); xh_a = (h - m_a) * r_a
        sc1, sh1 = mods[2 * i].chunk(2, -1); nx1 = xh_a * (1 + sc1) + sh1
        v = F.linear(nx1, Wv, bv); h_mid = h + F.linear(v, p[b + "attention.out_proj.weight"], p[b + "attention.out_proj.bias"])
        m_b, r_b = ln_stats(h_mid); xh_b = (h_mid - m_b) * r_b
        sc2, sh2 = mods[2 * i + 1].chunk(2, -1); nx2 = xh_b * (1 + sc2) + sh2
        a1 = F.linear(nx2, p[b + "mlp.0.weight"], p[b + "mlp.0.bias"]); u = F.gelu(a1)
        h_out = h_mid + F.linear(u, p[b + "mlp.2.weight"], p[b + "mlp.2.bias"])
        saved.append(dict(xh_a=xh_a, r_a=r_a, sc1=sc1, nx1=nx1, v=v, xh_b=xh_b, r_b=r_b, sc2=sc2, nx2=nx2, a1=a1, u=u))
        h = h_out
    m_f, r_f = ln_stats(h); xh_f = (h - m_f) * r_f
    scf, shf = mods[2 * nb].chunk(2, -1); nxf = xh_f * (1 + scf) + shf
    p0_pre = F.linear(nxf, p[S + "output_proj.0.weight"], p[S + "output_proj.0.bias"]); p0 = F.silu(p0_pre)
    raw = F.linear(p0, p[S + "output_proj.2.weight"])
    mult = p[S + "output_multiplier"]
    inside = ((raw >= -10) & (raw <= 10)).float()
    score = raw.clamp(-10, 10) * mult * tw
    # ---------------- tangent forward (only if z_dot given)
    tan = None
    if z_dot is not None:
        hd = F.linear(z_dot, p[S + "latent_proj.weight"])
        tan = []
        for i in range(nb):
            b = S + f"transformer_blocks.{i}."; sv = saved[i]
            Wv = p[b + "attention.in_proj_weight"][2 * D:]
            a_a = ln_T(sv["xh_a"], sv["r_a"], hd); nx1d = a_a * (1 + sv["sc1"])
            vd = F.linear(nx1d, Wv); hmd = hd + F.linear(vd, p[b + "attention.out_proj.weight"])
            a_b = ln_T(sv["xh_b"], sv["r_b"], hmd); nx2d = a_b * (1 + sv["sc2"])
            a1d = F.linear(nx2d, p[b + "mlp.0.weight"]); ud = gelu_d(sv["a1"]) * a1d
            hod = hmd + F.linear(ud, p[b + "mlp.2.weight"])
            tan.append(dict(hd_in=hd, a_a=a_a, nx1d=nx1d, vd=vd, hmd=hmd, a_b=a_b, nx2d=nx2d, a1d=a1d, ud=ud))
            hd = hod
        a_f = ln_T(xh_f, r_f, hd); nxfd = a_f * (1 + scf)
        p0pd = F.linear(nxfd, p[S + "output_proj.0.weight"]); p0d = silu_d(p0_pre) * p0pd
        rawd = F.linear(p0d, p[S + "output_proj.2.weight"])
        sd = inside * mult * tw * rawd
    # ---------------- reverse: lam (tangent adjoints), mu (primal adjoints)
    B = z.shape[0]
    if tan is None:
        mu_raw = d_score * mult * tw * inside
        acc(S + "output_multiplier", (d_score * raw.clamp(-10, 10) * tw).sum().reshape(1))
        lam_raw = None
    else:
        lam_raw = coef * inside * mult * tw
        acc(S + "output_multiplier", (coef * inside * tw * rawd).sum().reshape(1))
        mu_raw = torch.zeros_like(raw)
    dmods = [None] * (2 * nb + 1)
    def lin(name, W, x_in, mu_out, xd_in=None, lam_out=None, bias=True):
        gW = mu_out.t() @ x_in
        if lam_out is not None: gW = gW + lam_out.t() @ xd_in
        acc(name + ".weight", gW)
        if bias: acc(name + ".bias", mu_out.sum(0))
        return mu_out @ W, (lam_out @ W if lam_out is not None else None)
    mu_p0, lam_p0d = lin(S + "output_proj.2", p[S + "output_proj.2.weight"], p0, mu_raw, p0d if tan else None, lam_raw, bias=False)
    mu_p0pre = mu_p0 * silu_d(p0_pre)
    lam_p0pd = None
    if tan:
        lam_p0pd = lam_p0d * silu_d(p0_pre); mu_p0pre = mu_p0pre + lam_p0d * p0pd * silu_dd(p0_pre)
    mu_nxf, lam_nxfd = lin(S + "output_proj.0", p[S + "output_proj.0.weight"], nxf, mu_p0pre, nxfd if tan else None, lam_p0pd)
    def ada_bwd(idx, xh, r, scl, mu_nx, hd=None, a=None, lam_nxd=None):
        dscale = mu_nx * xh; dshift = mu_nx
        mu_h = ln_T(xh, r, mu_nx * (1 + scl))
        lam_hd = None
        if lam_nxd is not None:
            dscale = dscale + lam_nxd * a
            lam_a = lam_nxd * (1 + scl)
            lam_hd = ln_T(xh, r, lam_a)
            mu_h = mu_h + ln_second(xh, r, hd, lam_a)
        dmods[idx] = torch.cat([dscale, dshift], -1)
        return mu_h, lam_hd
    mu_h, lam_hd = ada_bwd(2 * nb, xh_f, r_f, scf, mu_nxf, hd if tan else None, a_f if tan else None, lam_nxfd)
    for i in reversed(range(nb)):
        b = S + f"transformer_blocks.{i}."; sv = saved[i]; tv_ = tan[i] if tan else None
        mu_u, lam_ud = lin(b + "mlp.2", p[b + "mlp.2.weight"], sv["u"], mu_h, tv_["ud"] if tan else None, lam_hd)
        mu_a1 = mu_u * gelu_d(sv["a1"]); lam_a1d = None
        if tan:
            lam_a1d = lam_ud * gelu_d(sv["a1"]); mu_a1 = mu_a1 + lam_ud * tv_["a1d"] * gelu_dd(sv["a1"])
        mu_nx2, lam_nx2d = lin(b + "mlp.0", p[b + "mlp.0.weight"], sv["nx2"], mu_a1, tv_["nx2d"] if tan else None, lam_a1d)
        dmu, dlam = ada_bwd(2 * i + 1, sv["xh_b"], sv["r_b"], sv["sc2"], mu_nx2, tv_["hmd"] if tan else None, tv_["a_b"] if tan else None, lam_nx2d)
        mu_hm = mu_h + dmu; lam_hmd = (lam_hd + dlam) if tan else None
        mu_v, lam_vd = lin(b + "attention.out_proj", p[b + "attention.out_proj.weight"], sv["v"], mu_hm, tv_["vd"] if tan else None, lam_hmd)
        Wv = p[b + "attention.in_proj_weight"][2 * D:]
        gW = mu_v.t() @ sv["nx1"] + (lam_vd.t() @ tv_["nx1d"] if tan else 0)
        full = torch.zeros_like(p[b + "attention.in_proj_weight"]); full[2 * D:] = gW; acc(b + "attention.in_proj_weight", full)
        fb = torch.zeros_like(p[b + "attention.in_proj_bias"]); fb[2 * D:] = mu_v.sum(0); acc(b + "attention.in_proj_bias", fb)
        mu_nx1 = mu_v @ Wv; lam_nx1d = lam_vd @ Wv if tan else None
        dmu, dlam = ada_bwd(2 * i, sv["xh_a"], sv["r_a"], sv["sc1"], mu_nx1, tv_["hd_in"] if tan else None, tv_["a_a"] if tan else None, lam_nx1d)
        mu_h = mu_hm + dmu; lam_hd = (lam_hmd + dlam) if tan else None
    acc(S + "latent_proj.weight", mu_h.t() @ z + (lam_hd.t() @ z_dot if tan else 0)); acc(S + "latent_proj.bias", mu_h.sum(0))
    dz = mu_h @ p[S + "latent_proj.weight"]
    # modulation -> cond path
    dsc = 0
    for i in range(2 * nb + 1):
        acc(ada_names(i) + ".weight", dmods[i].t() @ sc); acc(ada_names(i) + ".bias", dmods[i].sum(0))
        dsc = dsc + dmods[i] @ p[ada_names(i) + ".weight"]
    dcond = dsc * silu_d(cond)
    # obs path
    def ln_affine_bwd(dy, xh, r, gname, act_pre=None, msk=None):
        if msk is not None: dy = dy * msk
        if act_pre is not None: dy = dy * silu_d(act_pre)
        acc(gname + ".weight", (dy * xh).sum(0)); acc(gname + ".bias", dy.sum(0))
        return ln_T(xh, r, dy * p[gname + ".weight"])
    do2 = ln_affine_bwd(dcond, xh2, r2, S + "obs_encoder.8")
    acc(S + "obs_encoder.7.weight", do2.t() @ x2); acc(S + "obs_encoder.7.bias", do2.sum(0)); dx2 = do2 @ p[S + "obs_encoder.7.weight"]
    do1 = ln_affine_bwd(dx2, xh1, r1, S + "obs_encoder.5", y1)
    acc(S + "obs_encoder.4.weight", do1.t() @ x1); acc(S + "obs_encoder.4.bias", do1.sum(0)); dx1 = do1 @ p[S + "obs_encoder.4.weight"]
    do0 = ln_affine_bwd(dx1, xh0, r0, S + "obs_encoder.1", y0, mask)
    acc(S + "obs_encoder.0.weight", do0.t() @ obs); acc(S + "obs_encoder.0.bias", do0.sum(0))
    # time path
    dtemb = dcond
    acc(S + "time_scale", (dtemb * c2).sum().reshape(()))
    dc2 = ts * dtemb
    acc(S + "continuous_time_embed.4.weight", dc2.t() @ c1); acc(S + "continuous_time_embed.4.bias", dc2.sum(0))
    dc1p = (dc2 @ p[S + "continuous_time_embed.4.weight"]) * silu_d(c1_pre)
    acc(S + "continuous_time_embed.2.weight", dc1p.t() @ c0); acc(S + "continuous_time_embed.2.bias", dc1p.sum(0))
    dc0p = (dc1p @ p[S + "continuous_time_embed.2.weight"]) * silu_d(c0_pre)
    acc(S + "continuous_time_embed.0.weight", dc0p.t() @ cin); acc(S + "continuous_time_embed.0.bias", dc0p.sum(0))
    acc(S + "time_embed.3.weight", dtemb.t() @ te1); acc(S + "time_embed.3.bias", dtemb.sum(0))
    dte1p = (dtemb @ p[S + "time_embed.3.weight"]) * silu_d(te1_pre)
    acc(S + "time_embed.1.weight", dte1p.t() @ sin_e); acc(S + "time_embed.1.bias", dte1p.sum(0))
    dsin = dte1p @ p[S + "time_embed.1.weight"]
    acc(S + "time_embed.0.freq_scale", ((dsin[:, :half] * arg.cos() - dsin[:, half:] * arg.sin()) * (arg / fs)).sum().reshape(1))
    return G, dz, score, (sd if tan else None)


def main():
    torch.manual_seed(0)
    O, A, D, N, B = 17, 6, 64, 4, 8
    p = draw_params(param_specs(O, A, O, D, N), 5, orc.schedule_tables(N))
    ns = NoiseStream(3)
    z, obs, t = torch.from_numpy(ns.randn(B, O)), torch.from_numpy(ns.randn(B, O)), torch.from_numpy(ns.rand(B))
    mask = torch.from_numpy(ns.dropout_mask((B, D), 0.1)); dsc = torch.from_numpy(ns.randn(B, O))
    names = [n for n in p if n.startswith(S) and orc.is_param(n)]
    class MN:  # serves the fixed mask
        def dropout_mask(self, shape, pr): return mask.numpy()
    for n in names: p[n].requires_grad_(True)
    zz = z.clone().requires_grad_(True)
    s = orc.score_forward(p, zz, t, obs, MN(), True)
    ref = torch.autograd.grad((s * dsc).sum(), [p[n] for n in names] + [zz], allow_unused=True)
    with torch.no_grad():
        G, dz, score, _ = manual(p, z, t, obs, mask, d_score=dsc)
    worst = 0
    for n, g in zip(names, ref[:-1]):
        e = float((G[n] - g).abs().max()) / (float(g.abs().max()) + 1e-12); worst = max(worst, e)
        assert e < 2e-4, (n, e)
    assert float((dz - ref[-1]).abs().max()) < 1e-5
    print("main-path manual backward ok, worst rel err", worst)
    # gradient penalty
    zz = z.clone().requires_grad_(True)
    s = orc.score_forward(p, zz, t, obs, MN(), True)
    (g,) = torch.autograd.grad(s.sum(), zz, create_graph=True)
    pen = torch.mean((g.norm(2, dim=1) - 1.0) ** 2)
    ref = torch.autograd.grad(pen, [p[n] for n in names], allow_unused=True)
    with torch.no_grad():
        _, gman, _, _ = manual(p, z, t, obs, mask, d_score=torch.ones(B, O))
        nrm = gman.norm(2, dim=1, keepdim=True)
        w = (2.0 / B) * (nrm - 1) * gman / nrm
        G, _, _, sd = manual(p, z, t, obs, mask, z_dot=w, coef=1.0)
    assert float((gman - g.detach()).abs().max()) < 1e-5
    worst = 0
    for n, gr in zip(names, ref):
        if gr is None:
            continue
        e = float((G[n] - gr).abs().max()) / (float(gr.abs().max()) + 1e-12); worst = max(worst, e)
        assert e < 5e-4, (n, e, float(gr.abs().max()))
    print("gradient-penalty dual backward ok, worst rel err", worst, "penalty", float(pen))


if __name__ == "__main__":
    main()
