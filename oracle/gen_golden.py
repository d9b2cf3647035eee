#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE implementation (imported from /root/reference, CPU, fp32) on
seeded inputs and writes input/output fixtures to tests/golden/*.npz.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (the reference never travels to the GPU box);
the committed .npz files are data: inputs, recorded noise tensors and the reference's outputs.

    PYTHONPATH=/root/reference python3 oracle/gen_golden.py

The absent third-party wheels are stubbed in-process:
  * flash_attn.flash_attn_interface.flash_attn_varlen_qkvpacked_func -> exact fp32 softmax attention
    (non-causal, scale d^-1/2, equal-length sequences), the definition this build adopts for that boundary;
  * flash_attn.ops.fused_dense / omegaconf -> empty name holders (never called on this path).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("SEDD_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


# ----------------------------------------------------------------------------------------------- stubs
def _install_stubs():
    fa = types.ModuleType("flash_attn")
    fai = types.ModuleType("flash_attn.flash_attn_interface")

    def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None,
                                         causal=False, **kw):
        assert dropout_p == 0.0 and not causal
        total, three, H, d = qkv.shape
        B = cu_seqlens.numel() - 1
        L = total // B
        cu = [int(c) for c in cu_seqlens.tolist()]
        # a custom op: autocast only decides the dtype of what comes in, never the arithmetic inside (matters for g17,
        # where the caller's region runs under CPU autocast and einsum would otherwise be demoted to bf16)
        with torch.autocast("cpu", enabled=False):
            if cu != list(range(0, (B + 1) * L, L)) or B * L != total:
                # general variable-length form (flash_attn_interface.py:1212-1275): attention inside each segment
                # [cu[s], cu[s+1]) of the packed token axis; tokens outside every segment stay zero
                assert cu[0] == 0 and all(b >= a for a, b in zip(cu, cu[1:])) and cu[-1] <= total
                assert max(b - a for a, b in zip(cu, cu[1:])) <= max_seqlen
                out = torch.zeros(total, H, d, dtype=torch.float32)
                for a, b in zip(cu, cu[1:]):
                    if b > a:
                        q, k, v = qkv[a:b].float().unbind(1)
                        sc = torch.einsum("lhd,mhd->hlm", q, k) * (d ** -0.5)
                        out[a:b] = torch.einsum("hlm,mhd->lhd", torch.softmax(sc, dim=-1), v)
                return out.to(qkv.dtype)
            q, k, v = qkv.view(B, L, 3, H, d).float().unbind(2)
            s = torch.einsum("blhd,bmhd->bhlm", q, k) * (d ** -0.5)
            p = torch.softmax(s, dim=-1)
            o = torch.einsum("bhlm,bmhd->blhd", p, v)
            return o.reshape(total, H, d).to(qkv.dtype)

    fai.flash_attn_varlen_qkvpacked_func = flash_attn_varlen_qkvpacked_func
    ops = types.ModuleType("flash_attn.ops")
    fd = types.ModuleType("flash_attn.ops.fused_dense")
    fd.FusedMLP = object
    fd.FusedDense = object
    fa.flash_attn_interface = fai
    fa.ops = ops
    ops.fused_dense = fd
    sys.modules.update({"flash_attn": fa, "flash_attn.flash_attn_interface": fai, "flash_attn.ops": ops,
                        "flash_attn.ops.fused_dense": fd})
    oc = types.ModuleType("omegaconf")

    class OmegaConf:  # noqa
        @staticmethod
        def create(d):
            return d

    oc.OmegaConf = OmegaConf
    sys.modules["omegaconf"] = oc


_install_stubs()

import builtins  # noqa: E402

_real_print = builtins.print

from protlig_dd.processing import graph_lib, noise_lib, losses  # noqa: E402
from protlig_dd.sampling import sampling  # noqa: E402
from protlig_dd.utils import catsample  # noqa: E402
from protlig_dd.model import transformer as ref_tf  # noqa: E402
from protlig_dd.model import transformers_protlig_cross as ref_cross  # noqa: E402
from protlig_dd.model import rotary as ref_rotary  # noqa: E402
from protlig_dd.model.ema import ExponentialMovingAverage  # noqa: E402

from oracle.weights import load_seeded  # noqa: E402

NS = types.SimpleNamespace


def quiet(fn, *a, **k):
    """The reference prints shapes inside forward(); silence it while generating."""
    builtins.print = lambda *aa, **kk: None
    try:
        return fn(*a, **k)
    finally:
        builtins.print = _real_print


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    _real_print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def sedd_config(hidden, heads, cond, blocks, tokens=25, graph="absorb", dropout=0.0, scale_by_sigma=True):
    return NS(tokens=tokens, graph=NS(type=graph),
              model=NS(hidden_size=hidden, n_heads=heads, cond_dim=cond, n_blocks=blocks, dropout=dropout,
                       scale_by_sigma=scale_by_sigma, length=128))


class RecordRand:
    """Context manager: records every torch.rand / rand_like / randint_like the reference draws."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._rand, self._rand_like, self._randint_like = torch.rand, torch.rand_like, torch.randint_like
        log = self.log

        def rand(*a, **k):
            r = self._rand(*a, **k)
            log.append(("rand", r.clone()))
            return r

        def rand_like(*a, **k):
            r = self._rand_like(*a, **k)
            log.append(("rand_like", r.clone()))
            return r

        def randint_like(*a, **k):
            r = self._randint_like(*a, **k)
            log.append(("randint_like", r.clone()))
            return r

        torch.rand, torch.rand_like, torch.randint_like = rand, rand_like, randint_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like, torch.randint_like = self._rand, self._rand_like, self._randint_like


# ----------------------------------------------------------------------------------------------- G1
def g1_noise():
    t = torch.cat([torch.linspace(1e-3, 1.0, 61), torch.tensor([1e-5, 0.5, 0.999])])
    out = {"t": t}
    ll = noise_lib.LogLinearNoise(eps=1e-3)
    out["loglinear_sigma"], out["loglinear_dsigma"] = ll(t)
    ll2 = noise_lib.LogLinearNoise(eps=0.1)
    out["loglinear01_sigma"], out["loglinear01_dsigma"] = ll2(t)
    ge = noise_lib.GeometricNoise(1e-4, 20.0)
    out["geometric_sigma"], out["geometric_dsigma"] = ge(t)
    co = noise_lib.CosineNoise(1e-3, 1.0)
    out["cosine_sigma"], out["cosine_dsigma"] = co(t)
    save("g1_noise", **out)


# ----------------------------------------------------------------------------------------------- G2
def g2_sample_transition():
    torch.manual_seed(102)
    out = {}
    B, L = 6, 40
    sigma = torch.tensor([1e-3, 0.05, 0.4, 0.7, 2.0, 6.9])[:, None]
    out["sigma"] = sigma
    for V, tag in ((26, "absorb26"), (2365, "absorb2365")):
        g = graph_lib.Absorbing(V - 1)
        x0 = torch.randint(0, V - 1, (B, L))
        with RecordRand() as rr:
            xt = g.sample_transition(x0, sigma)
        out[tag + "_x0"], out[tag + "_u"], out[tag + "_xt"] = x0, rr.log[0][1], xt
    g = graph_lib.Uniform(25)
    x0 = torch.randint(0, 25, (B, L))
    with RecordRand() as rr:
        xt = g.sample_transition(x0, sigma)
    out["uniform25_x0"], out["uniform25_u"], out["uniform25_r"], out["uniform25_xt"] = \
        x0, rr.log[0][1], rr.log[1][1], xt
    # curriculum variant (graph_lib.py:236-256) at a few training steps
    g = graph_lib.Absorbing(25)
    x0 = torch.randint(0, 25, (B, L))
    out["curr_x0"] = x0
    steps = [0, 99, 100, 300, 500, 999, 1000, 1200, 1500, 1999, 2000, 5000]
    out["curr_steps"] = np.array(steps)
    us, xts = [], []
    for s in steps:
        with RecordRand() as rr:
            xt = g.sample_transition_curriculum(x0, sigma, s)
        us.append(rr.log[0][1])
        xts.append(xt)
    out["curr_u"], out["curr_xt"] = torch.stack(us), torch.stack(xts)
    save("g2_sample_transition", **out)


# ----------------------------------------------------------------------------------------------- G3
def g3_score_entropy():
    torch.manual_seed(103)
    out = {}
    for V, B, L in ((26, 4, 24), (37, 4, 24), (2365, 2, 8)):
        g = graph_lib.Absorbing(V - 1)
        sigma = torch.tensor([0.02, 0.3, 0.8, 5.0])[:B, None]
        x0 = torch.randint(0, V - 1, (B, L))
        x = torch.where(torch.rand(B, L) < 0.6, torch.full_like(x0, V - 1), x0)
        x[0, :] = V - 1          # fully masked row
        if B > 2:
            x[1, :] = x0[1, :]   # fully unmasked row
        score = (torch.randn(B, L, V) * 1.5).requires_grad_(True)
        w = torch.randn(B, L)
        se = g.score_entropy(score, sigma, x, x0)
        (se * w).sum().backward()
        t = f"absorb{V}_"
        out.update({t + "sigma": sigma, t + "x": x, t + "x0": x0, t + "score": score, t + "w": w,
                    t + "out": se, t + "grad": score.grad})
    V, B, L = 25, 4, 24
    g = graph_lib.Uniform(V)
    sigma = torch.tensor([0.02, 0.3, 0.8, 5.0])[:, None]
    x0 = torch.randint(0, V, (B, L))
    x = torch.where(torch.rand(B, L) < 0.5, torch.randint(0, V, (B, L)), x0)
    score = (torch.randn(B, L, V) * 1.5).requires_grad_(True)
    w = torch.randn(B, L)
    se = g.score_entropy(score, sigma, x, x0)
    (se * w).sum().backward()
    t = "uniform25_"
    out.update({t + "sigma": sigma, t + "x": x, t + "x0": x0, t + "score": score, t + "w": w,
                t + "out": se, t + "grad": score.grad})
    save("g3_score_entropy", **out)


# ----------------------------------------------------------------------------------------------- G4
def g4_graph_ops():
    torch.manual_seed(104)
    out = {}
    B, L = 3, 20
    for tag, g, V in (("absorb26", graph_lib.Absorbing(25), 26), ("uniform25", graph_lib.Uniform(25), 25)):
        i = torch.randint(0, V, (B, L))
        if g.absorb:
            i[:, ::3] = V - 1
        score = torch.randn(B, L, V).exp()
        sigma = torch.tensor([0.03, 0.6, 3.0])[:, None]
        t = tag + "_"
        out[t + "i"], out[t + "score"], out[t + "sigma"] = i, score, sigma
        out[t + "rate"] = g.rate(i)
        out[t + "transp_rate"] = g.transp_rate(i)
        out[t + "reverse_rate"] = g.reverse_rate(i, score)
        out[t + "staggered_score"] = g.staggered_score(score, sigma)
        out[t + "transp_transition"] = g.transp_transition(i, sigma)
        if not g.absorb:
            out[t + "transition"] = g.transition(i, sigma)
        with RecordRand() as rr:
            lim = g.sample_limit(B, L)
        out[t + "sample_limit"] = lim
        rev = 0.01 * sigma[..., None] * g.reverse_rate(i, score)
        with RecordRand() as rr:
            nxt = g.sample_rate(i, rev)
        out[t + "sample_rate_rate"], out[t + "sample_rate_u"], out[t + "sample_rate_out"] = rev, rr.log[0][1], nxt
    save("g4_graph_ops", **out)


# ----------------------------------------------------------------------------------------------- G5
def g5_sample_categorical():
    torch.manual_seed(105)
    out = {}
    for V, n in ((26, 512), (37, 256), (2365, 64)):
        p = torch.rand(n, V) ** 3
        p[::7] = 0.0
        p[::7, 3] = 1.0                      # one-hot rows
        p[1::11] = p[1::11].round()          # rows with many exact ties (0/1 values)
        with RecordRand() as rr:
            s = catsample.sample_categorical(p)
        out[f"v{V}_p"], out[f"v{V}_u"], out[f"v{V}_out"] = p, rr.log[0][1], s
    save("g5_sample_categorical", **out)


# ----------------------------------------------------------------------------------------------- G6
class _FakeScoreModel:
    """Returns fixed log-scores; accepts the Euler/Denoiser keyword signature (sampling.py:65-76)."""

    def __init__(self, logits):
        self.logits = logits
        self.sigmas = []

    def eval(self):
        return self

    def __call__(self, **kw):
        self.sigmas.append(kw["timesteps"].clone())
        return self.logits


def g6_predictor_steps():
    torch.manual_seed(106)
    out = {}
    for V, B, L, tag in ((26, 4, 32, "v26"), (2365, 2, 8, "v2365")):
        graph = graph_lib.Absorbing(V - 1)
        noise = noise_lib.LogLinearNoise(eps=1e-3)
        x = torch.randint(0, V - 1, (B, L))
        x = torch.where(torch.rand(B, L) < 0.7, torch.full_like(x, V - 1), x)
        logits = torch.randn(B, L, V)
        logits.scatter_(-1, x[..., None], torch.zeros(B, L, 1))
        t = torch.tensor([0.9, 0.5, 0.2, 0.05])[:B, None]
        dt = 1.0 / 16
        out[tag + "_x"], out[tag + "_logits"], out[tag + "_t"], out[tag + "_dt"] = x, logits, t, np.float64(dt)
        # Euler
        m = _FakeScoreModel(logits)
        with RecordRand() as rr:
            xe = quiet(sampling.EulerPredictor(graph, noise).update_fn, score_model=m, x=x, t=t, step_size=dt,
                       task="protein_only")
        out[tag + "_euler_u"], out[tag + "_euler_out"], out[tag + "_sigma_seen"] = rr.log[0][1], xe, m.sigmas[0]
        # Analytic (upstream positional signature): score_fn returns exp(logits)
        with RecordRand() as rr:
            xa = sampling.AnalyticPredictor(graph, noise).update_fn(lambda xx, s: logits.exp(), x, t, dt)
        out[tag + "_analytic_u"], out[tag + "_analytic_out"] = rr.log[0][1], xa
        # Denoiser at t = eps
        te = torch.full((B, 1), 1e-5)
        m = _FakeScoreModel(logits)
        with RecordRand() as rr:
            xd = sampling.Denoiser(graph, noise).update_fn(score_model=m, x=x, t=te, step_size=dt, task="protein_only")
        out[tag + "_denoise_t"], out[tag + "_denoise_u"], out[tag + "_denoise_out"] = te, rr.log[0][1], xd
    # Uniform graph, Euler + Analytic + Denoiser
    V, B, L = 25, 4, 32
    graph = graph_lib.Uniform(V)
    noise = noise_lib.GeometricNoise(1e-4, 20.0)
    x = torch.randint(0, V, (B, L))
    logits = torch.randn(B, L, V)
    logits.scatter_(-1, x[..., None], torch.zeros(B, L, 1))
    t = torch.tensor([0.9, 0.5, 0.2, 0.05])[:, None]
    dt = 1.0 / 16
    tag = "u25"
    out[tag + "_x"], out[tag + "_logits"], out[tag + "_t"], out[tag + "_dt"] = x, logits, t, np.float64(dt)
    m = _FakeScoreModel(logits)
    with RecordRand() as rr:
        xe = quiet(sampling.EulerPredictor(graph, noise).update_fn, score_model=m, x=x, t=t, step_size=dt,
                   task="protein_only")
    out[tag + "_euler_u"], out[tag + "_euler_out"] = rr.log[0][1], xe
    with RecordRand() as rr:
        xa = sampling.AnalyticPredictor(graph, noise).update_fn(lambda xx, s: logits.exp(), x, t, dt)
    out[tag + "_analytic_u"], out[tag + "_analytic_out"] = rr.log[0][1], xa
    te = torch.full((B, 1), 1e-5)
    m = _FakeScoreModel(logits)
    with RecordRand() as rr:
        xd = sampling.Denoiser(graph, noise).update_fn(score_model=m, x=x, t=te, step_size=dt, task="protein_only")
    out[tag + "_denoise_t"], out[tag + "_denoise_u"], out[tag + "_denoise_out"] = te, rr.log[0][1], xd
    save("g6_predictor_steps", **out)


# ----------------------------------------------------------------------------------------------- G7
def g7_timestep_embed():
    torch.manual_seed(107)
    emb = ref_tf.TimestepEmbedder(128)
    load_seeded(emb, seed=7)
    sigma = torch.tensor([1e-3, 0.01, 0.3, 0.69, 2.5, 6.9, 20.0])
    freq = emb.timestep_embedding(sigma, 256)
    c = F.silu(emb(sigma))
    w = torch.randn_like(c)
    (c * w).sum().backward()
    out = {"sigma": sigma, "freq": freq, "c": c, "w": w}
    for n, p in emb.named_parameters():
        out["grad." + n] = p.grad
    save("g7_timestep_embed", **out)


# ----------------------------------------------------------------------------------------------- G8
def g8_ddit_block():
    torch.manual_seed(108)
    out = {}
    for tag, D, H, C, B, L in (("d32", 128, 4, 64, 3, 48), ("d64", 128, 2, 64, 2, 80), ("d96", 192, 2, 64, 2, 40)):
        blk = ref_tf.DDiTBlock(D, H, C, dropout=0.0)
        load_seeded(blk, seed=8, std=0.05)
        blk.eval()
        x = torch.randn(B, L, D, requires_grad=True)
        c = torch.randn(B, C, requires_grad=True)
        rot = ref_rotary.Rotary(D // H)
        cos, sin = rot(x)
        y = blk(x, (cos, sin), c)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        t = f"flash_{tag}_"
        out.update({t + "x": x, t + "c": c, t + "y": y, t + "w": w, t + "dx": x.grad, t + "dc": c.grad,
                    t + "dims": np.array([D, H, C, B, L])})
        for n, p in blk.named_parameters():
            out[t + "grad." + n] = p.grad
    # cross variant (transformers_protlig_cross.py:237-320): batch-axis self-attention quirk Q1 as written,
    # with and without other_track, with and without key-padding mask
    D, H, C, B, L, Lo = 64, 2, 32, 3, 20, 12
    blk = ref_cross.DDiTBlock(D, H, 0.0, C)
    load_seeded(blk, seed=9, std=0.05)
    blk.eval()
    x = torch.randn(B, L, D, requires_grad=True)
    c = torch.randn(B, C, requires_grad=True)
    other = torch.randn(B, Lo, D, requires_grad=True)
    mask = torch.ones(B, Lo, dtype=torch.bool)
    mask[0, 9:] = False
    mask[2, 5:] = False
    rot = ref_rotary.Rotary(D // H)
    cos, sin = rot(x)
    t = "cross_"
    out.update({t + "x": x, t + "c": c, t + "other": other, t + "mask": mask, t + "dims": np.array([D, H, C, B, L, Lo])})
    y0 = blk(x, (cos, sin), c)
    out[t + "y_noother"] = y0
    y1 = blk(x, (cos, sin), c, other_track=other)
    w = torch.randn_like(y1)
    out[t + "w"] = w
    (y1 * w).sum().backward()
    out.update({t + "y": y1, t + "dx": x.grad.clone(), t + "dc": c.grad.clone(), t + "dother": other.grad.clone()})
    for n, p in blk.named_parameters():
        out[t + "grad." + n] = p.grad.clone()
    y2 = blk(x, (cos, sin), c, other_track=other, other_mask=mask)
    out[t + "y_masked"] = y2
    save("g8_ddit_block", **out)


# ----------------------------------------------------------------------------------------------- G9
def g9_cross_attention():
    torch.manual_seed(109)
    D, H, B, Lq, Lk = 64, 2, 3, 20, 12
    layer = ref_cross.CrossAttentionLayer(D, H, dropout=0.0)
    load_seeded(layer, seed=10, std=0.08)
    layer.eval()
    x = torch.randn(B, Lq, D, requires_grad=True)
    ctx = torch.randn(B, Lk, D, requires_grad=True)
    mask = torch.ones(B, Lk, dtype=torch.bool)
    mask[1, 7:] = False
    out = {"x": x, "ctx": ctx, "mask": mask, "dims": np.array([D, H, B, Lq, Lk])}
    y = layer(x, ctx)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    out.update({"y": y, "w": w, "dx": x.grad.clone(), "dctx": ctx.grad.clone()})
    for n, p in layer.named_parameters():
        out["grad." + n] = p.grad.clone()
        p.grad = None
    x.grad = None
    ctx.grad = None
    ym = layer(x, ctx, context_mask=mask)
    (ym * w).sum().backward()
    out.update({"y_masked": ym, "dx_masked": x.grad.clone(), "dctx_masked": ctx.grad.clone()})
    save("g9_cross_attention", **out)


# ----------------------------------------------------------------------------------------------- G10
class _KwDrop(torch.nn.Module):
    """losses.get_loss_fn -> model/utils.py:36 passes prot_seq=/lig_seq=/task= that SEDD.forward does not take."""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x, sigma, **kw):
        return self.m(x, sigma)


def _loss_inputs(B, L, V, seed):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randint(0, V - 1, (B, L), generator=g)
    t = (1 - 1e-3) * torch.rand(B, generator=g) + 1e-3
    u = torch.rand(B, L, generator=g)
    return x0, t, u


def g10_sedd_model():
    out = {}
    graph = graph_lib.Absorbing(25)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    for tag, cfg, B, L in (("mini", sedd_config(128, 4, 64, 2), 3, 64),
                           ("mini96", sedd_config(192, 2, 64, 2), 2, 40),
                           ("tiny", sedd_config(768, 8, 128, 8), 8, 128)):
        model = ref_tf._SEDD(cfg)
        load_seeded(model, seed=10)
        model.eval()
        x0, t, u = _loss_inputs(B, L, 26, 1000 + len(tag))
        sigma, dsigma = noise(t)
        xt = torch.where(u < 1 - (-sigma[:, None]).exp(), torch.full_like(x0, 25), x0)
        logits = quiet(model, xt, sigma)
        loss_fn = losses.get_loss_fn(noise, graph, train=False)
        loss = quiet(loss_fn, _KwDrop(model), x0, t=t, perturbed_batch=xt)
        t_ = tag + "_"
        out.update({t_ + "x0": x0, t_ + "t": t, t_ + "u": u, t_ + "xt": xt, t_ + "sigma": sigma,
                    t_ + "logits": logits, t_ + "loss": loss,
                    t_ + "cfg": np.array([cfg.model.hidden_size, cfg.model.n_heads, cfg.model.cond_dim,
                                          cfg.model.n_blocks, cfg.tokens])})
        if tag != "tiny":
            model.zero_grad()
            loss.backward()
            for n, p in model.named_parameters():
                out[t_ + "grad." + n] = p.grad
    # KAT-init-loss: untouched reference init -> logits = -log(e^sigma-1) - log(V-1)
    cfg = sedd_config(128, 4, 64, 2)
    model = ref_tf._SEDD(cfg).eval()
    x0, t, u = _loss_inputs(4, 32, 26, 77)
    sigma, dsigma = noise(t)
    xt = torch.where(u < 1 - (-sigma[:, None]).exp(), torch.full_like(x0, 25), x0)
    out["init_x0"], out["init_t"], out["init_xt"] = x0, t, xt
    out["init_logits"] = quiet(model, xt, sigma)
    out["init_loss"] = quiet(losses.get_loss_fn(noise, graph, train=False), _KwDrop(model), x0, t=t, perturbed_batch=xt)
    save("g10_sedd_model", **out)


# ----------------------------------------------------------------------------------------------- G11
def cross_config(hidden, heads, cond, blocks, dropout=0.0):
    return NS(graph=NS(type="absorb"),
              model=NS(hidden_size=hidden, n_heads=heads, cond_dim=cond, n_blocks=blocks, dropout=dropout,
                       scale_by_sigma=True, device="cpu"))


def g11_cross_model():
    torch.manual_seed(111)
    cfg = cross_config(64, 2, 32, 2)
    model = ref_cross.ProteinLigandDiffusionModel(cfg)
    load_seeded(model, seed=11, std=0.05)
    model.eval()
    B, Lp, Ll = 3, 24, 16
    gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    p0 = torch.randint(0, 33, (B, Lp))
    l0 = torch.randint(0, 2362, (B, Ll))
    t = torch.tensor([0.15, 0.6, 0.93])
    sigma, dsigma = noise(t)
    up, ul = torch.rand(B, Lp), torch.rand(B, Ll)
    move = 1 - (-sigma[:, None]).exp()
    pt = torch.where(up < move, torch.full_like(p0, 36), p0)
    lt = torch.where(ul < move, torch.full_like(l0, 2364), l0)
    out = {"p0": p0, "l0": l0, "t": t, "up": up, "ul": ul, "pt": pt, "lt": lt, "sigma": sigma, "dsigma": dsigma,
           "cfg": np.array([64, 2, 32, 2])}
    lp, ll = model(protein_indices=pt, ligand_indices=lt, timesteps=sigma, mode="joint")
    out["joint_prot_logits"], out["joint_lig_logits"] = lp, ll
    # calc_loss joint (run_train_protlig_cross_att.py:284-294): Q3 as written, and the per-sample form
    se_p = gp.score_entropy(lp, sigma[:, None], pt, p0)
    se_l = gl.score_entropy(ll, sigma[:, None], lt, l0)
    loss_q3 = (dsigma[:, None] * ((se_p.mean() + se_l.mean()) / 2)).mean()
    loss_ps = ((dsigma[:, None] * se_p).mean() + (dsigma[:, None] * se_l).mean()) / 2
    out["joint_loss_q3"], out["joint_loss_persample"] = loss_q3, loss_ps
    model.zero_grad()
    loss_q3.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["joint_grad." + n] = p.grad.clone()
    out["prot_only_logits"] = model(protein_indices=pt, timesteps=sigma, mode="protein_only")
    out["lig_only_logits"] = model(ligand_indices=lt, timesteps=sigma, mode="ligand_only")
    out["lig_given_prot_logits"] = model(protein_indices=p0, ligand_indices=lt, timesteps=sigma,
                                         mode="ligand_given_protein")
    out["prot_given_lig_logits"] = model(protein_indices=pt, ligand_indices=l0, timesteps=sigma,
                                         mode="protein_given_ligand")
    save("g11_cross_model", **out)


# ----------------------------------------------------------------------------------------------- G12
def g12_train_trajectory():
    cfg = sedd_config(128, 4, 64, 2)
    model = ref_tf._SEDD(cfg)
    load_seeded(model, seed=12)
    graph = graph_lib.Absorbing(25)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    params = list(model.parameters()) + list(noise.parameters())
    opt = torch.optim.AdamW(params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.9999)
    loss_fn = losses.get_loss_fn(noise, graph, train=True)
    wrapped = _KwDrop(model)
    B, L, steps, warmup, lr = 4, 64, 4, 2, 3e-4
    out = {"cfg": np.array([128, 4, 64, 2, 25]), "hyper": np.array([lr, 0.9, 0.999, 1e-8, 0.0, warmup, 1.0, 0.9999])}
    ls, gns = [], []
    for s in range(steps):
        x0, t, u = _loss_inputs(B, L, 26, 500 + s)
        sigma, _ = noise(t)
        xt = torch.where(u < 1 - (-sigma[:, None]).exp(), torch.full_like(x0, 25), x0)
        out[f"x0_{s}"], out[f"t_{s}"], out[f"xt_{s}"] = x0, t, xt
        loss = quiet(loss_fn, wrapped, x0, t=t, perturbed_batch=xt)
        opt.zero_grad()
        loss.backward()
        # optimize_fn (losses.py:60-69) without the GradScaler (fp32, scale == 1)
        for g in opt.param_groups:
            g["lr"] = lr * np.minimum((s + 1) / warmup, 1.0)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        ema.update(model.parameters())
        ls.append(loss.detach())
        gns.append(gn.detach())
    out["losses"], out["grad_norms"] = torch.stack(ls), torch.stack(gns)
    sd = model.state_dict()
    out["final.blocks.0.attn_qkv.weight"] = sd["blocks.0.attn_qkv.weight"]
    out["final.output_layer.linear.bias"] = sd["output_layer.linear.bias"]
    out["final.vocab_embed.embedding.weight"] = sd["vocab_embed.embedding.weight"]
    names = [n for n, _ in model.named_parameters()]
    i = names.index("blocks.0.attn_qkv.weight")
    out["ema.blocks.0.attn_qkv.weight"] = ema.shadow_params[i]
    out["ema_num_updates"] = np.array(ema.num_updates)
    save("g12_train_trajectory", **out)


# ----------------------------------------------------------------------------------------------- G13
class _SamplerModel(torch.nn.Module):
    """Adapts mini _SEDD to the keyword call of EulerPredictor/Denoiser (sampling.py:65-76,111-122)."""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, protein_indices=None, ligand_indices=None, timesteps=None, mode=None):
        return self.m(protein_indices, timesteps)


def g13_pc_sampler():
    cfg = sedd_config(128, 4, 64, 2)
    model = ref_tf._SEDD(cfg)
    load_seeded(model, seed=13, std=0.1)
    graph = graph_lib.Absorbing(25)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    B, L, steps = 3, 32, 6
    out = {"cfg": np.array([128, 4, 64, 2, 25]), "dims": np.array([B, L, steps])}
    torch.manual_seed(113)
    sampler = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, denoise=True, eps=1e-5)
    with RecordRand() as rr:
        x = quiet(sampler, _SamplerModel(model), "protein_only")
    out["euler_out"] = x
    out["euler_u"] = torch.stack([r for _, r in rr.log[:steps]])
    out["euler_u_denoise"] = rr.log[steps][1]
    # Analytic predictor through the same loop (upstream signature; the reference loop cannot call it, Q5):
    # replicate get_pc_sampler's loop (sampling.py:157-173) around AnalyticPredictor.update_fn / Denoiser.
    torch.manual_seed(114)
    pred = sampling.AnalyticPredictor(graph, noise)
    den = sampling.Denoiser(graph, noise)
    model.eval()
    score_fn = lambda xx, s: quiet(model, xx, s.reshape(-1)).exp()  # noqa: E731
    xs = []
    with torch.no_grad(), RecordRand() as rr:
        x = graph.sample_limit(B, L)
        timesteps = torch.linspace(1, 1e-5, steps + 1)
        dt = (1 - 1e-5) / steps
        for i in range(steps):
            t = timesteps[i] * torch.ones(B, 1)
            x = pred.update_fn(score_fn, x, t, dt)
            xs.append(x.clone())
        t = timesteps[-1] * torch.ones(B, 1)
        x = quiet(den.update_fn, score_model=_SamplerModel(model), x=x, t=t, step_size=dt, task="protein_only")
    out["analytic_out"] = x
    out["analytic_steps"] = torch.stack(xs)
    out["analytic_u"] = torch.stack([r for _, r in rr.log[:steps]])
    out["analytic_u_denoise"] = rr.log[steps][1]
    save("g13_pc_sampler", **out)


# ----------------------------------------------------------------------------------------------- G14
def g14_ema():
    torch.manual_seed(114)
    lin = torch.nn.Linear(5, 7)
    ema = ExponentialMovingAverage(lin.parameters(), decay=0.999)
    out = {"w0": lin.weight.detach().clone(), "b0": lin.bias.detach().clone()}
    ws, bs = [], []
    for s in range(12):
        with torch.no_grad():
            for p in lin.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        ws.append(lin.weight.detach().clone())
        bs.append(lin.bias.detach().clone())
        ema.update(lin.parameters())
    out["ws"], out["bs"] = torch.stack(ws), torch.stack(bs)
    out["shadow_w"], out["shadow_b"] = ema.shadow_params[0], ema.shadow_params[1]
    out["num_updates"] = np.array(ema.num_updates)
    save("g14_ema", **out)


def g15_lr_schedule():
    """utils/lr_scheduler.py:51-87 WarmupCosineLR stepped through warm-up, decay and past max_steps."""
    from protlig_dd.utils.lr_scheduler import WarmupCosineLR
    out = {}
    for tag, kw in (("a", dict(warmup_steps=10, max_steps=50, base_lr=1e-5, max_lr=1e-3, min_lr=1e-6)),
                    ("b", dict(warmup_steps=0, max_steps=7, base_lr=3e-4, max_lr=3e-4, min_lr=0.0))):
        lin = torch.nn.Linear(2, 2)
        opt = torch.optim.SGD(lin.parameters(), lr=0.1)
        sch = WarmupCosineLR(opt, **kw)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(kw["max_steps"] + 5):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        out[tag + "_lrs"] = np.array(lrs, dtype=np.float64)
        out[tag + "_cfg"] = np.array([kw["warmup_steps"], kw["max_steps"], kw["base_lr"], kw["max_lr"], kw["min_lr"]], dtype=np.float64)
    save("g15_lr_schedule", **out)


def g16_cross_train_trajectory():
    """run_train_protlig_cross_att.py:244-333 — `calc_loss` + `train_step` followed line by line on the reference's
    model / graphs / noise / EMA / WarmupCosineLR (the Trainer class itself imports wandb and the dataset stack, so its
    step is replayed here): joint task, quirks as written (Q2: optimizer stepped twice per iteration, Q3: joint loss
    = mean(dsigma) * (mean(SE_prot) + mean(SE_lig)) / 2), GradScaler omitted (fp32: scale 1), dropout 0."""
    from protlig_dd.utils.lr_scheduler import WarmupCosineLR
    torch.manual_seed(116)
    cfg = cross_config(64, 2, 32, 2)
    model = ref_cross.ProteinLigandDiffusionModel(cfg)
    load_seeded(model, seed=16, std=0.05)
    model.train()
    gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    opt = torch.optim.AdamW(list(model.parameters()) + list(noise.parameters()), lr=3e-4, betas=(0.9, 0.999),
                            eps=1e-8, weight_decay=0.01)
    sched = WarmupCosineLR(opt, warmup_steps=2, max_steps=10, base_lr=1e-5, max_lr=1e-3, min_lr=1e-6)
    B, Lp, Ll, iters = 3, 24, 16, 3
    out = {"cfg": np.array([64, 2, 32, 2]), "hyper": np.array([0.9, 0.999, 1e-8, 0.01, 1.0, 0.999, 2, 10, 1e-5, 1e-3, 1e-6])}
    ls, lps, lls, lrs = [], [], [], []
    for it in range(iters):
        g = torch.Generator().manual_seed(1600 + it)
        p0 = torch.randint(0, 33, (B, Lp), generator=g)
        l0 = torch.randint(0, 2362, (B, Ll), generator=g)
        t = (1 - 1e-3) * torch.rand(B, generator=g) + 1e-3
        up, ul = torch.rand(B, Lp, generator=g), torch.rand(B, Ll, generator=g)
        sigma, dsigma = noise(t)
        move = 1 - (-sigma[:, None]).exp()
        pt = torch.where(up < move, torch.full_like(p0, 36), p0)   # graph.sample_transition with recorded uniforms
        lt = torch.where(ul < move, torch.full_like(l0, 2364), l0)
        out[f"p0_{it}"], out[f"l0_{it}"], out[f"t_{it}"], out[f"pt_{it}"], out[f"lt_{it}"] = p0, l0, t, pt, lt
        lrs.append(opt.param_groups[0]["lr"])
        log_score = model(protein_indices=pt, ligand_indices=lt, timesteps=sigma.reshape(-1), mode="joint")
        loss_prot = gp.score_entropy(log_score[0], sigma[:, None], pt, p0).mean()
        loss_lig = gl.score_entropy(log_score[1], sigma[:, None], lt, l0).mean()
        loss = (loss_prot + loss_lig) / 2
        loss = (dsigma[:, None] * loss).mean()
        loss_prot = (dsigma[:, None] * loss_prot).mean()
        loss_lig = (dsigma[:, None] * loss_lig).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)   # optimize_fn :244-249
        opt.step()                                                 # scaler.step(optimizer)
        ema.update(model.parameters())                             # :324
        opt.step()                                                 # :325 (Q2)
        opt.zero_grad()
        sched.step()
        ls.append(loss.detach()); lps.append(loss_prot.detach()); lls.append(loss_lig.detach())
    out["losses"], out["losses_prot"], out["losses_lig"] = torch.stack(ls), torch.stack(lps), torch.stack(lls)
    out["lrs"] = np.array(lrs, dtype=np.float64)
    sd = model.state_dict()
    names = [n for n, _ in model.named_parameters()]
    for key in ("transformer.blocks_prot.0.attn_qkv.weight", "transformer.blocks_lig.1.mlp.2.bias",
                "transformer.ligand_output.linear.weight", "transformer.protein_embed.embedding.weight"):
        out["final." + key] = sd[key]
        out["ema." + key] = ema.shadow_params[names.index(key)]
    out["ema_num_updates"] = np.array(ema.num_updates)
    save("g16_cross_train_trajectory", **out)


# ----------------------------------------------------------------------------------------------- G17
class _CpuBf16Autocast:
    """Runs the reference's own `torch.cuda.amp.autocast(...)` regions (transformer.py:36, 169, 306-311) as CPU autocast:
    without CUDA the reference's contexts are inert and its forward is plain fp32, so the bf16 rounding points of the
    autocast region would otherwise be pinned by nothing but this build's restatement.  `torch.cuda.amp.autocast` is
    looked up at call time, so swapping the attribute for the duration of a forward reroutes every use — including the
    `enabled=False` islands around LayerNorm and rotary — with no change to the reference's code."""

    def __enter__(self):
        self.saved = torch.cuda.amp.autocast
        torch.cuda.amp.autocast = lambda dtype=torch.bfloat16, enabled=True, **kw: torch.autocast(
            "cpu", dtype=dtype or torch.bfloat16, enabled=enabled)
        return self

    def __exit__(self, *a):
        torch.cuda.amp.autocast = self.saved
        return False


def g17_bf16_autocast():
    """The reference under bf16 autocast (CPU autocast standing in for CUDA autocast; both cast nn.Linear / matmul
    operands and results to bf16 and leave the explicitly-disabled islands in fp32): one flash-variant DDiTBlock
    (forward + gradients, entered under the same autocast context SEDD.forward opens, transformer.py:306) and the
    mini `_SEDD` (logits, loss, gradients).  Known difference between the two autocast op lists that matters here: none
    on this path — every op between the Linears (modulate, GELU, gate/residual) is elementwise and runs in its input
    dtype under either backend; LayerNorm and rotary run inside the reference's own `enabled=False` islands."""
    torch.manual_seed(117)
    out = {}
    D, H, C, B, L = 128, 2, 64, 2, 80
    blk = ref_tf.DDiTBlock(D, H, C, dropout=0.0)
    load_seeded(blk, seed=8, std=0.05)
    blk.eval()
    x = torch.randn(B, L, D, requires_grad=True)
    c = torch.randn(B, C, requires_grad=True)
    rot = ref_rotary.Rotary(D // H)
    cos, sin = rot(x)
    with _CpuBf16Autocast():
        with torch.cuda.amp.autocast(dtype=torch.bfloat16):
            y = blk(x, (cos, sin), c)
    w = torch.randn(B, L, D)
    (y.float() * w).sum().backward()
    t = "block_"
    out.update({t + "x": x, t + "c": c, t + "y": y.float(), t + "w": w, t + "dx": x.grad, t + "dc": c.grad,
                t + "dims": np.array([D, H, C, B, L]), t + "y_is_fp32": np.array(int(y.dtype == torch.float32))})
    for n, p in blk.named_parameters():
        out[t + "grad." + n] = p.grad
    graph = graph_lib.Absorbing(25)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    cfg, B, L = sedd_config(128, 4, 64, 2), 3, 64
    model = ref_tf._SEDD(cfg)
    load_seeded(model, seed=10)
    model.eval()
    x0, tt, u = _loss_inputs(B, L, 26, 1017)
    sigma, dsigma = noise(tt)
    xt = torch.where(u < 1 - (-sigma[:, None]).exp(), torch.full_like(x0, 25), x0)
    with _CpuBf16Autocast():
        logits = quiet(model, xt, sigma)
        loss = quiet(losses.get_loss_fn(noise, graph, train=False), _KwDrop(model), x0, t=tt, perturbed_batch=xt)
    model.zero_grad()
    loss.backward()
    t = "mini_"
    out.update({t + "x0": x0, t + "t": tt, t + "xt": xt, t + "sigma": sigma, t + "logits": logits.float(),
                t + "logits_is_bf16": np.array(int(logits.dtype == torch.bfloat16)), t + "loss": loss.float(),
                t + "cfg": np.array([cfg.model.hidden_size, cfg.model.n_heads, cfg.model.cond_dim, cfg.model.n_blocks,
                                     cfg.tokens])})
    for n, p in model.named_parameters():
        out[t + "grad." + n] = p.grad
    save("g17_bf16_autocast", **out)


# ----------------------------------------------------------------------------------------------- G18
def g18_conditional_sampling():
    """the two conditional-sampling front-ends, run through the reference's own get_pc_sampler loop with the
    noise it draws recorded.
      (a) infilling (sampling/run_sample_cond.py:32-44): `proj_fun` clamps a prefix and a suffix before every update, the
          caller applies it once more to the result;
      (b) `ligand_given_protein` / `protein_given_ligand` on the dual-track model (sampling.py:64-76: the conditioning
          track's indices ride along, transformers_protlig_cross.py:596-642 builds the cross-attention context)."""
    out = {}
    cfg = sedd_config(128, 4, 64, 2)
    model = ref_tf._SEDD(cfg)
    load_seeded(model, seed=13, std=0.1)
    graph = graph_lib.Absorbing(25)
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    B, L, steps = 3, 40, 6
    gen = torch.Generator().manual_seed(118)
    prefix = torch.randint(0, 25, (5,), generator=gen)
    suffix = torch.randint(0, 25, (7,), generator=gen)
    input_locs = list(range(len(prefix))) + list(range(L - len(suffix), L))
    input_ids = torch.cat([prefix, suffix])[None].repeat(B, 1)

    def proj_fun(x):
        x[:, input_locs] = input_ids
        return x

    for pred_name in ("euler",):
        torch.manual_seed(118)
        sampler = sampling.get_pc_sampler(graph, noise, (B, L), pred_name, steps, denoise=True, eps=1e-5, proj_fun=proj_fun)
        with RecordRand() as rr:
            x = proj_fun(quiet(sampler, _SamplerModel(model), "protein_only"))
        out["infill_out"] = x
        out["infill_u"] = torch.stack([r for _, r in rr.log[:steps]])
        out["infill_u_denoise"] = rr.log[steps][1]
    out.update({"infill_cfg": np.array([128, 4, 64, 2, 25]), "infill_dims": np.array([B, L, steps]),
                "infill_locs": np.array(input_locs), "infill_ids": input_ids})
    # (b) conditional tracks of the dual-track model
    ccfg = cross_config(64, 2, 32, 2)
    cm = ref_cross.ProteinLigandDiffusionModel(ccfg)
    load_seeded(cm, seed=11, std=0.05)
    cm.eval()
    B, Lp, Ll, steps = 3, 24, 16, 5
    gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
    prot = torch.randint(0, 33, (B, Lp), generator=gen)
    lig = torch.randint(0, 2362, (B, Ll), generator=gen)
    out.update({"cond_cfg": np.array([64, 2, 32, 2]), "cond_dims": np.array([B, Lp, Ll, steps]), "cond_prot": prot,
                "cond_lig": lig})
    torch.manual_seed(218)
    sampler = sampling.get_pc_sampler(gl, noise, (B, Ll), "euler", steps, denoise=True, eps=1e-5)
    with RecordRand() as rr:
        x = quiet(sampler, cm, "ligand_given_protein", prot, None)
    out["lgp_out"], out["lgp_u"], out["lgp_u_denoise"] = x, torch.stack([r for _, r in rr.log[:steps]]), rr.log[steps][1]
    torch.manual_seed(318)
    sampler = sampling.get_pc_sampler(gp, noise, (B, Lp), "euler", steps, denoise=True, eps=1e-5)
    with RecordRand() as rr:
        x = quiet(sampler, cm, "protein_given_ligand", None, lig)
    out["pgl_out"], out["pgl_u"], out["pgl_u_denoise"] = x, torch.stack([r for _, r in rr.log[:steps]]), rr.log[steps][1]
    save("g18_conditional_sampling", **out)


# ----------------------------------------------------------------------------------------------- G19
def g19_varlen_block():
    """the flash-attn varlen hook of DDiTBlock.forward (transformer.py:173-179) with UNEQUAL segment lengths:
    `seqlens` starts with 0 (cu_seqlens = seqlens.cumsum(-1) must start at 0 for flash-attn), segments of different
    lengths that also straddle the padded rows (variable-length packing), forward + all gradients through the stub."""
    torch.manual_seed(119)
    out = {}
    for tag, D, H, C, B, L, lens in (("a", 128, 2, 64, 2, 80, [0, 33, 47, 50, 30]),       # 4 segments over 2 x 80 tokens
                                     ("b", 128, 4, 64, 3, 48, [0, 48, 7, 41, 20, 28])):   # incl. a 7-token segment
        assert sum(lens) == B * L and max(lens) <= L
        blk = ref_tf.DDiTBlock(D, H, C, dropout=0.0)
        load_seeded(blk, seed=19, std=0.05)
        blk.eval()
        x = torch.randn(B, L, D, requires_grad=True)
        c = torch.randn(B, C, requires_grad=True)
        rot = ref_rotary.Rotary(D // H)
        cos, sin = rot(x)
        seqlens = torch.tensor(lens, dtype=torch.int32)
        y = blk(x, (cos, sin), c, seqlens=seqlens)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        t = tag + "_"
        out.update({t + "x": x, t + "c": c, t + "y": y, t + "w": w, t + "dx": x.grad, t + "dc": c.grad,
                    t + "seqlens": seqlens, t + "dims": np.array([D, H, C, B, L])})
        for n, p in blk.named_parameters():
            out[t + "grad." + n] = p.grad
        # the hook changes the result: same block, one segment per row
        out[t + "y_rows"] = blk(x, (cos, sin), c)
    save("g19_varlen_block", **out)


# ----------------------------------------------------------------------------------------------- G20
def sedd_cond_config(hidden, heads, cond, blocks, tokens, esm_dim, mol_dim):
    cfg = sedd_config(hidden, heads, cond, blocks, tokens=tokens)
    cfg.model.esm_dim, cfg.model.molformer_dim = esm_dim, mol_dim
    return cfg


def _load_sedd_cond(model, seed):
    """seeded weights + BatchNorm scales moved to 1 + N(0, 0.02) and non-trivial running statistics (the tests apply
    the same recipe by name)."""
    load_seeded(model, seed=seed)
    with torch.no_grad():
        for bn in (model.esm_norm, model.mol_norm):
            bn.weight.add_(1.0)
        g = torch.Generator().manual_seed(seed + 7)
        for bn in (model.esm_norm, model.mol_norm):
            bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.05)
            bn.running_var.copy_(1.0 + 0.3 * torch.rand(bn.running_var.shape, generator=g))


def g20_sedd_conditioned():
    """`SEDD` (transformer.py:240-325) — the class north_star names — WITH its pooled conditioners: esm_proj / mol_proj +
    BatchNorm1d + mean over [c, esm_c, mol_c] (:289-301), as the DDP trainer calls it (run_train_ddp_protlig_transf.py:284
    passes esm_cond / mol_cond).  Cases: 'train' (BatchNorm batch statistics + running-stat update; logits, loss, all
    gradients), 'eval' (running statistics), 'esm' (mol_cond=None: the two-term mean), and 'large' = the model dims of
    configs/config_large.yaml with its 2396-token vocabulary at L 1024 (2 blocks; logits stored on a row subsample)."""
    out = {}
    noise = noise_lib.LogLinearNoise(eps=1e-3)
    # ---- mini: every branch, with gradients
    cfg = sedd_cond_config(128, 4, 64, 2, 25, 40, 48)
    graph = graph_lib.Absorbing(25)
    model = ref_tf.SEDD(cfg)
    _load_sedd_cond(model, seed=20)
    B, L = 5, 48
    x0, t, u = _loss_inputs(B, L, 26, 2020)
    g = torch.Generator().manual_seed(2021)
    esm, mol = torch.randn(B, 40, generator=g), torch.randn(B, 48, generator=g) * 0.7 + 0.1
    sigma, dsigma = noise(t)
    xt = torch.where(u < 1 - (-sigma[:, None]).exp(), torch.full_like(x0, 25), x0)
    out.update({"mini_x0": x0, "mini_t": t, "mini_xt": xt, "mini_sigma": sigma, "mini_esm": esm, "mini_mol": mol,
                "mini_cfg": np.array([128, 4, 64, 2, 25, 40, 48])})
    for bn, nm in ((model.esm_norm, "esm_norm"), (model.mol_norm, "mol_norm")):
        out[f"mini_{nm}.running_mean0"], out[f"mini_{nm}.running_var0"] = bn.running_mean.clone(), bn.running_var.clone()
    model.eval()
    out["mini_eval_logits"] = quiet(model, xt, sigma, esm, mol)
    out["mini_esm_eval_logits"] = quiet(model, xt, sigma, esm, None)
    out["mini_none_eval_logits"] = quiet(model, xt, sigma)
    model.train()
    logits = quiet(model, xt, sigma, esm, mol)
    se = graph.score_entropy(logits, sigma[:, None], xt, x0)
    loss = (dsigma[:, None] * se).mean()
    model.zero_grad()
    loss.backward()
    out["mini_train_logits"], out["mini_train_loss"] = logits, loss
    for n, p in model.named_parameters():
        out["mini_train_grad." + n] = p.grad
    for bn, nm in ((model.esm_norm, "esm_norm"), (model.mol_norm, "mol_norm")):
        out[f"mini_{nm}.running_mean1"], out[f"mini_{nm}.running_var1"] = bn.running_mean.clone(), bn.running_var.clone()
        out[f"mini_{nm}.num_batches_tracked1"] = bn.num_batches_tracked.clone()
    # a second training forward with only the ESM conditioner: the molecule BatchNorm must stay untouched
    logits2 = quiet(model, xt, sigma, esm, None)
    out["mini_esm_train_logits"] = logits2
    out["mini_esm_norm.running_mean2"] = model.esm_norm.running_mean.clone()
    out["mini_mol_norm.running_mean2"] = model.mol_norm.running_mean.clone()
    # ---- large: configs/config_large.yaml dims (D 1024, 16 heads, cond 128, tokens 2396, L 1024, esm 640 / mol 768), 2 blocks
    cfg = sedd_cond_config(1024, 16, 128, 2, 2396, 640, 768)
    graph = graph_lib.Absorbing(2396)
    model = ref_tf.SEDD(cfg)
    _load_sedd_cond(model, seed=24)
    model.eval()
    B, L = 2, 1024
    x0, t, u = _loss_inputs(B, L, 2397, 2024)
    g = torch.Generator().manual_seed(2025)
    esm, mol = torch.randn(B, 640, generator=g), torch.randn(B, 768, generator=g)
    sigma, dsigma = noise(t)
    xt = torch.where(u < 1 - (-sigma[:, None]).exp(), torch.full_like(x0, 2396), x0)
    with torch.no_grad():
        logits = quiet(model, xt, sigma, esm, mol)
        se = graph.score_entropy(logits, sigma[:, None], xt, x0)
        loss = (dsigma[:, None] * se).mean()
    rows = torch.arange(0, L, 37)
    out.update({"large_x0": x0, "large_t": t, "large_xt": xt, "large_sigma": sigma, "large_esm": esm, "large_mol": mol,
                "large_cfg": np.array([1024, 16, 128, 2, 2396, 640, 768]), "large_rows": rows,
                "large_logits_rows": logits[:, rows], "large_loss": loss, "large_se": se,
                "large_logits_rowsum": logits.double().sum(-1).float()})
    save("g20_sedd_conditioned", **out)


# ----------------------------------------------------------------------------------------------- G21
def g21_plinder_collate():
    """the reference's PLINDER list dataset + collate (data/process_full_plinder.py:31-243) and the seeded
    split / DistributedSampler shards of its loaders (:293-301, :396-398) on a SYNTHETIC sample list (strings of random
    lengths incl. over-long ones, coordinates in both layouts, binding values, interaction masks).  Stored: the sample
    list itself (strings as unicode arrays), the collated batch, the split and the per-rank index orders."""
    import tempfile
    from protlig_dd.data import process_full_plinder as pfp
    from torch.utils.data import DistributedSampler, random_split
    rng = np.random.RandomState(21)
    aa, sm = "ACDEFGHIKLMNPQRSTVWY", "CNOc1()=#SFl[]@H+-23"
    out = {}
    for tag, n, coords3 in (("a", 7, True), ("b", 5, False)):
        samples = []
        for i in range(n):
            lp, ll = int(rng.randint(5, 90)), int(rng.randint(3, 40))
            s = {"protein_seq": "".join(rng.choice(list(aa), lp)), "ligand_smiles": "".join(rng.choice(list(sm), ll)),
                 "protein_coords": rng.randn(min(lp, 64), 3, 3).astype(np.float32) if coords3 else rng.randn(min(lp, 64), 3).astype(np.float32),
                 "ligand_coords": rng.randn(int(rng.randint(2, 30)), 3).astype(np.float32),
                 "binding_value": float(rng.randn()), "interaction_mask": (rng.rand(min(lp, 64)) > 0.5).astype(np.float32)}
            samples.append(s)
        path = os.path.join(tempfile.mkdtemp(), "synthetic.pt")
        torch.save(samples, path)
        ds = pfp.ImprovedProteinLigandDataset(path, max_protein_len=64, max_ligand_len=24)
        batch = pfp.improved_protein_ligand_collate_fn([ds[i] for i in range(n)])
        t = tag + "_"
        out[t + "protein_seq"] = np.array([s["protein_seq"] for s in samples])
        out[t + "ligand_smiles"] = np.array([s["ligand_smiles"] for s in samples])
        out[t + "binding_in"] = np.array([s["binding_value"] for s in samples], dtype=np.float64)
        for i, s in enumerate(samples):
            out[f"{t}pc_{i}"], out[f"{t}lc_{i}"], out[f"{t}im_{i}"] = s["protein_coords"], s["ligand_coords"], s["interaction_mask"]
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                out[t + "batch_" + k] = v
        assert batch["protein_seq"] == [s["protein_seq"] for s in samples]
    # a batch without the optional fields and with only the ligand side
    ds = pfp.ImprovedProteinLigandDataset.__new__(pfp.ImprovedProteinLigandDataset)
    ds.data, ds.max_protein_len, ds.max_ligand_len = [{"ligand_smiles": "CCO"}, {"ligand_smiles": "c1ccccc1O"}], 16, 8
    b = pfp.improved_protein_ligand_collate_fn([ds[0], ds[1]])
    out["ligonly_keys"] = np.array(sorted(b.keys()))
    out["ligonly_mask"] = b["ligand_mask"]
    # split + shards
    for n in (10, 37):
        parts = random_split(list(range(n)), [int(0.8 * n), int(0.1 * n), n - int(0.8 * n) - int(0.1 * n)],
                             generator=torch.Generator().manual_seed(42))
        for nm, p in zip(("train", "val", "test"), parts):
            out[f"split{n}_{nm}"] = np.array(list(p.indices), dtype=np.int64)
        for world in (2, 3):
            for rank in range(world):
                sp = DistributedSampler(parts[0], num_replicas=world, rank=rank)
                sp.set_epoch(2)
                out[f"shard{n}_w{world}_r{rank}"] = np.array([parts[0].indices[i] for i in sp], dtype=np.int64)
    save("g21_plinder_collate", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for fn in (g1_noise, g2_sample_transition, g3_score_entropy, g4_graph_ops, g5_sample_categorical,
               g6_predictor_steps, g7_timestep_embed, g8_ddit_block, g9_cross_attention, g10_sedd_model,
               g11_cross_model, g12_train_trajectory, g13_pc_sampler, g14_ema, g15_lr_schedule,
               g16_cross_train_trajectory, g17_bf16_autocast, g18_conditional_sampling,
               g19_varlen_block, g20_sedd_conditioned, g21_plinder_collate):
        if only and not any(fn.__name__.startswith(o) for o in only):
            continue
        fn()


if __name__ == "__main__":
    main()
