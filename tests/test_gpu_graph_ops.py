"""GPU parity: HIP graph/loss/sampling kernels (through the C ABI) vs the committed golden vectors captured from the
reference and vs the CPU oracle on seeded inputs.  Bar: sampled indices bit-exact given identical noise; fp32
loss within 1e-4 relative; bf16 inputs 2e-2."""
import math

import pytest
import torch

from conftest import assert_close
from oracle import sedd_cpu as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _graphs():
    from protein_lig_sedd_amd import graph_lib
    return graph_lib


def _mismatch(a, b):
    return int((a.cpu() != b.cpu()).sum())


def test_g2_sample_transition(golden):
    GL = _graphs()
    g = golden("g2_sample_transition")
    sigma = g["sigma"]
    for V in (26, 2365):
        t = f"absorb{V}_"
        out = GL.Absorbing(V - 1).sample_transition(g[t + "x0"].to(DEV), sigma.to(DEV), u=g[t + "u"].to(DEV))
        assert _mismatch(out, g[t + "xt"]) == 0
    out = GL.Uniform(25).sample_transition(g["uniform25_x0"].to(DEV), sigma.to(DEV), u=g["uniform25_u"].to(DEV),
                                           r=g["uniform25_r"].to(DEV))
    assert _mismatch(out, g["uniform25_xt"]) == 0
    gr = GL.Absorbing(25)
    for k, s in enumerate(g["curr_steps"].tolist()):
        out = gr.sample_transition_curriculum(g["curr_x0"].to(DEV), sigma.to(DEV), s, u=g["curr_u"][k].to(DEV))
        assert _mismatch(out, g["curr_xt"][k]) == 0, f"curriculum step {s}"


def test_g3_score_entropy(golden):
    GL = _graphs()
    g = golden("g3_score_entropy")
    for tag, graph in (("absorb26_", GL.Absorbing(25)), ("absorb37_", GL.Absorbing(36)),
                       ("absorb2365_", GL.Absorbing(2364)), ("uniform25_", GL.Uniform(25))):
        score = g[tag + "score"].to(DEV).requires_grad_(True)
        se = graph.score_entropy(score, g[tag + "sigma"].to(DEV), g[tag + "x"].to(DEV), g[tag + "x0"].to(DEV))
        (se * g[tag + "w"].to(DEV)).sum().backward()
        assert_close(se, g[tag + "out"], 1e-4, 1e-5, tag + "fwd")
        assert_close(score.grad, g[tag + "grad"], 1e-4, 1e-6, tag + "grad")


def test_score_entropy_fused_and_bf16():
    """fused loss path (gw) == autograd path; bf16 scores within 2e-2 of the fp32 oracle."""
    from protein_lig_sedd_amd import ops
    GL = _graphs()
    torch.manual_seed(0)
    for graph, og, V in ((GL.Absorbing(25), O.AbsorbingCPU(25), 26), (GL.Uniform(40), O.UniformCPU(40), 40),
                         (GL.Absorbing(70), O.AbsorbingCPU(70), 71)):
        B, L = 6, 50
        score = torch.randn(B, L, V)
        sigma = torch.rand(B) * 3 + 0.05
        x0 = torch.randint(0, V - 1, (B, L))
        x = torch.where(torch.rand(B, L) < 0.5, torch.full_like(x0, V - 1) if og.absorb else torch.randint(0, V, (B, L)), x0)
        gw = torch.rand(B)
        ref_s = score.clone().requires_grad_(True)
        ref = og.score_entropy(ref_s, sigma[:, None], x, x0)
        (ref * gw[:, None]).sum().backward()
        out, ds = ops.score_entropy_fwd(graph, score.to(DEV), sigma.to(DEV), x.to(DEV), x0.to(DEV), gw=gw.to(DEV))
        assert_close(out, ref, 1e-4, 1e-5)
        assert_close(ds, ref_s.grad, 1e-4, 1e-6)
        sb = score.to(torch.bfloat16)
        refb = og.score_entropy(sb.float(), sigma[:, None], x, x0)
        outb, dsb = ops.score_entropy_fwd(graph, sb.to(DEV), sigma.to(DEV), x.to(DEV), x0.to(DEV), gw=gw.to(DEV))
        assert outb.dtype == torch.float32 and dsb.dtype == torch.bfloat16
        assert_close(outb, refb, 1e-4, 1e-5)
        assert_close(dsb.float(), ref_s.grad, 2e-2, 2e-2)


def test_score_entropy_edges():
    from protein_lig_sedd_amd import ops
    GL = _graphs()
    graph = GL.Absorbing(25)
    # empty batch
    out, _ = ops.score_entropy_fwd(graph, torch.empty(0, 8, 26, device=DEV), torch.empty(0, device=DEV),
                                   torch.empty(0, 8, dtype=torch.int64, device=DEV),
                                   torch.empty(0, 8, dtype=torch.int64, device=DEV))
    assert out.shape == (0, 8)
    # x0 out of range is clamped (graph_lib.py:289); all-unmasked rows give exactly 0
    score = torch.randn(2, 4, 26)
    x = torch.full((2, 4), 25)
    x0 = torch.tensor([[30, -3, 0, 25]] * 2)
    sigma = torch.tensor([0.3, 2.0])
    ref = O.AbsorbingCPU(25).score_entropy(score, sigma[:, None], x, x0)
    out, _ = ops.score_entropy_fwd(graph, score.to(DEV), sigma.to(DEV), x.to(DEV), x0.to(DEV))
    assert_close(out, ref, 1e-4, 1e-5)
    out, _ = ops.score_entropy_fwd(graph, score.to(DEV), sigma.to(DEV), torch.zeros(2, 4, dtype=torch.int64, device=DEV),
                                   x0.clamp(0, 25).to(DEV))
    assert float(out.abs().max()) == 0.0


def test_g4_graph_methods(golden):
    GL = _graphs()
    g = golden("g4_graph_ops")
    for tag, graph in (("absorb26_", GL.Absorbing(25)), ("uniform25_", GL.Uniform(25))):
        i, score, sigma = g[tag + "i"].to(DEV), g[tag + "score"].to(DEV), g[tag + "sigma"].to(DEV)
        assert_close(graph.rate(i), g[tag + "rate"], 1e-6, 0)
        assert_close(graph.transp_rate(i), g[tag + "transp_rate"], 1e-6, 0)
        assert_close(graph.reverse_rate(i, score), g[tag + "reverse_rate"], 1e-5, 1e-6)
        assert_close(graph.staggered_score(score, sigma), g[tag + "staggered_score"], 1e-5, 1e-6)
        assert_close(graph.transp_transition(i, sigma), g[tag + "transp_transition"], 1e-6, 1e-8)
        out = graph.sample_rate(i, g[tag + "sample_rate_rate"].to(DEV), u=g[tag + "sample_rate_u"].to(DEV))
        assert _mismatch(out, g[tag + "sample_rate_out"]) == 0
    assert torch.equal(GL.Absorbing(25).sample_limit(3, 20), g["absorb26_sample_limit"])
    assert GL.Absorbing(25).transp_rate(g["absorb26_i"].to(DEV)).dtype == torch.int64


def test_g5_sample_categorical(golden):
    from protein_lig_sedd_amd.catsample import sample_categorical
    g = golden("g5_sample_categorical")
    for V in (26, 37, 2365):
        out = sample_categorical(g[f"v{V}_p"].to(DEV), u=g[f"v{V}_u"].to(DEV))
        assert _mismatch(out, g[f"v{V}_out"]) == 0, f"V={V}"
    with pytest.raises(ValueError):
        sample_categorical(g["v26_p"].to(DEV), method="soft")


def test_g6_predictor_steps(golden):
    from protein_lig_sedd_amd import _lib as L
    GL = _graphs()
    g = golden("g6_predictor_steps")
    noise = O.make_noise("loglinear", eps=1e-3)
    cases = [("v26", GL.Absorbing(25), noise), ("v2365", GL.Absorbing(2364), noise),
             ("u25", GL.Uniform(25), O.make_noise("geometric", sigma_min=1e-4, sigma_max=20.0))]
    for tag, graph, nz in cases:
        x, logits, t, dt = g[tag + "_x"].to(DEV), g[tag + "_logits"].to(DEV), g[tag + "_t"], float(g[tag + "_dt"])
        sig, dsig = nz(t)
        out = graph.predictor_step(L.STEP_EULER, logits, x, (dt * dsig).to(DEV), u=g[tag + "_euler_u"].to(DEV))
        assert _mismatch(out, g[tag + "_euler_out"]) == 0, tag + " euler"
        ds = sig - nz(t - dt)[0]
        out = graph.predictor_step(L.STEP_ANALYTIC, logits, x, ds.to(DEV), u=g[tag + "_analytic_u"].to(DEV))
        assert _mismatch(out, g[tag + "_analytic_out"]) == 0, tag + " analytic"
        sg = nz(g[tag + "_denoise_t"])[0]
        out = graph.predictor_step(L.STEP_DENOISE, logits, x, sg.to(DEV), u=g[tag + "_denoise_u"].to(DEV))
        assert _mismatch(out, g[tag + "_denoise_out"]) == 0, tag + " denoise"


@pytest.mark.parametrize("V", [8, 9, 26, 37, 64, 65, 300])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_predictor_steps_vs_oracle(V, dtype):
    """ragged vocab sizes across all group widths, both graphs, bf16 and fp32 log-scores, seeded."""
    from protein_lig_sedd_amd import _lib as L
    GL = _graphs()
    torch.manual_seed(V)
    B, Ln = 5, 77
    noise = O.make_noise("loglinear", eps=1e-3)
    for graph, og in ((GL.Absorbing(V - 1), O.AbsorbingCPU(V - 1)), (GL.Uniform(V), O.UniformCPU(V))):
        x = torch.randint(0, V, (B, Ln))
        if og.absorb:
            x[torch.rand(B, Ln) < 0.5] = V - 1
        logits = (torch.randn(B, Ln, V) * 2 - 3).to(dtype)
        logits = logits.scatter(-1, x[..., None], torch.zeros(B, Ln, 1, dtype=dtype))
        t = torch.rand(B, 1) * 0.9 + 0.05
        dt = 0.01
        sig, dsig = noise(t)
        u = torch.rand(B, Ln, V)
        ref = O.euler_step(og, logits, x, (sig, dsig), dt, u)
        out = graph.predictor_step(L.STEP_EULER, logits.to(DEV), x.to(DEV), (dt * dsig).to(DEV), u=u.to(DEV))
        assert _mismatch(out, ref) == 0, "euler"
        ds = sig - noise(t - dt)[0]
        ref = O.analytic_step(og, logits, x, ds, u)
        out = graph.predictor_step(L.STEP_ANALYTIC, logits.to(DEV), x.to(DEV), ds.to(DEV), u=u.to(DEV))
        assert _mismatch(out, ref) == 0, "analytic"
        ud = u[..., :-1].contiguous() if og.absorb else u
        ref = O.denoise_step(og, logits, x, sig, ud)
        out = graph.predictor_step(L.STEP_DENOISE, logits.to(DEV), x.to(DEV), sig.to(DEV), u=ud.to(DEV))
        assert _mismatch(out, ref) == 0, "denoise"
        if og.absorb:
            assert int((out == V - 1).sum()) == 0  # the denoiser never leaves a mask


@pytest.mark.parametrize("V", [9, 26, 37, 63, 64, 300])  # 64 and 300: the philox_uniform + explicit-noise route
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_predictor_step_philox_matches_explicit_noise_and_oracle(V, dtype):
    """sedd_predictor_step_philox = sedd_predictor_step on the uniforms the header defines (oracle.philox_uniform: the
    published Philox4x32-10, pinned by the Random123 known answers in the CPU suite) = the oracle's step; bit-exact."""
    from protein_lig_sedd_amd import _lib as L, ops
    GL = _graphs()
    torch.manual_seed(100 + V)
    B, Ln = 3, 171  # ragged last tile of rows
    noise = O.make_noise("loglinear", eps=1e-3)
    seed, offset = 0x9E3779B97F4A7C15, 41  # a seed above 2^63: passes through the signed binding intact
    for graph, og in ((GL.Absorbing(V - 1), O.AbsorbingCPU(V - 1)), (GL.Uniform(V), O.UniformCPU(V))):
        x = torch.randint(0, V, (B, Ln))
        if og.absorb:
            x[torch.rand(B, Ln) < 0.6] = V - 1
        logits = (torch.randn(B, Ln, V) * 2 - 3).to(dtype)
        t = torch.rand(B, 1) * 0.9 + 0.05
        sig, dsig = noise(t)
        dt = 0.01
        ds = sig - noise(t - dt)[0]
        u = O.philox_uniform(seed, offset, B * Ln, V).view(B, Ln, V)
        ud = O.philox_uniform(seed, offset, B * Ln, V - 1).view(B, Ln, V - 1) if og.absorb else u
        cases = ((L.STEP_EULER, dt * dsig, u, O.euler_step(og, logits, x, (sig, dsig), dt, u)),
                 (L.STEP_ANALYTIC, ds, u, O.analytic_step(og, logits, x, ds, u)),
                 (L.STEP_DENOISE, sig, ud, O.denoise_step(og, logits, x, sig, ud)))
        for step, coef, uu, ref in cases:
            out = graph.predictor_step(step, logits.to(DEV), x.to(DEV), coef.to(DEV), u=ops.PhiloxNoise(seed, offset))
            exp = graph.predictor_step(step, logits.to(DEV), x.to(DEV), coef.to(DEV), u=uu.to(DEV))
            assert _mismatch(out, exp) == 0, f"step {step}: in-kernel noise vs the same noise passed in"
            assert _mismatch(out, ref) == 0, f"step {step}: vs oracle"
        other = graph.predictor_step(L.STEP_ANALYTIC, logits.to(DEV), x.to(DEV), ds.to(DEV), u=ops.PhiloxNoise(seed, offset + 1))
        assert _mismatch(other, cases[1][3]) > 0  # another offset is another draw


def test_pc_sampler_with_philox_stream_equals_explicit_noise_sequence():
    """get_pc_sampler(noise_seq=PhiloxStream(seed)) = the same sampler fed the tensors philox_uniform defines."""
    from protein_lig_sedd_amd import ops, sampling, noise_lib
    GL = _graphs()
    V, B, Ln, steps = 26, 4, 96, 5
    graph = GL.Absorbing(V - 1)
    noise = noise_lib.LogLinearNoise()
    torch.manual_seed(5)
    W = torch.randn(V, V, device=DEV) * 0.5

    class Model(torch.nn.Module):  # a stand-in score model: log-score depends on the token and the noise level
        def forward(self, x, sigma):
            return (W[x] - sigma.view(-1, 1, 1)).bfloat16()

    model = Model()

    for name in ("euler", "analytic"):
        fn = sampling.get_pc_sampler(graph, noise, (B, Ln), name, steps, denoise=True, device=torch.device(DEV))
        a = fn(model, noise_seq=ops.PhiloxStream(77))
        seq = [O.philox_uniform(77, i, B * Ln, V if i < steps else V - 1).view(B, Ln, -1).to(DEV) for i in range(steps + 1)]
        b = fn(model, noise_seq=seq)
        assert _mismatch(a, b) == 0, name
        assert int((a == V - 1).sum()) == 0


def _misaligned(t):
    """same values at an address that is not 16-byte aligned (routes the C ABI to the lane-group kernels)."""
    buf = torch.empty(t.numel() + 8, dtype=t.dtype, device=t.device)
    v = buf[1:1 + t.numel()].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 != 0 and v.is_contiguous()
    return v


@pytest.mark.parametrize("V", [9, 26, 37, 50])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_row_per_lane_layout_matches_group_layout(V, dtype):
    """V <= 64 has two kernel layouts (LDS-staged row-per-lane for aligned buffers, lane-group otherwise): same bits
    for loss, gradient and sampled ids; sizes cover several tiles, a ragged last tile and a <16-byte tail."""
    from protein_lig_sedd_amd import ops, _lib as L
    GL = _graphs()
    torch.manual_seed(100 + V)
    B, Ln = 7, 301
    for graph in (GL.Absorbing(V - 1), GL.Uniform(V)):
        absorb = isinstance(graph, GL.Absorbing)
        x0 = torch.randint(0, V - 1, (B, Ln), device=DEV)
        x = torch.randint(0, V, (B, Ln), device=DEV)
        if absorb:
            x[torch.rand(B, Ln, device=DEV) < 0.5] = V - 1
        score = (torch.randn(B, Ln, V, device=DEV) * 2 - 1).to(dtype)
        sigma = torch.rand(B, device=DEV) * 3 + 0.05
        gw = torch.rand(B, device=DEV)
        u = torch.rand(B, Ln, V, device=DEV)
        o1, d1 = ops.score_entropy_fwd(graph, score, sigma, x, x0, gw=gw)
        o2, d2 = ops.score_entropy_fwd(graph, _misaligned(score), sigma, x, x0, gw=gw)
        assert torch.equal(o1, o2) and torch.equal(d1, d2)
        assert torch.equal(ops.gumbel_argmax(score.exp(), u), ops.gumbel_argmax(_misaligned(score.exp()), u))
        for step in (L.STEP_EULER, L.STEP_ANALYTIC, L.STEP_DENOISE):
            uu = u[..., :-1].contiguous() if (absorb and step == L.STEP_DENOISE) else u
            a = ops.predictor_step(graph, step, score, x, sigma, uu)
            b = ops.predictor_step(graph, step, _misaligned(score), x, sigma, uu)
            assert torch.equal(a, b), (type(graph).__name__, step)


def test_absorbing_unmasked_fixed_point_full_size():
    """SURVEY section 4 KAT at BASELINE config-5 shard size (512 x 1024 rows): Euler/Analytic never move an
    unmasked token, whatever the noise; masked tokens only move to non-mask ids or stay."""
    from protein_lig_sedd_amd import _lib as L
    GL = _graphs()
    torch.manual_seed(1)
    graph = GL.Absorbing(25)
    B, Ln, V = 512, 1024, 26
    x = torch.randint(0, V, (B, Ln), device=DEV)
    logits = (torch.randn(B, Ln, V, device=DEV) - 2).to(torch.bfloat16)
    logits.scatter_(-1, x[..., None], 0.0)
    coef = torch.full((B,), 0.05, device=DEV)
    for step in (L.STEP_EULER, L.STEP_ANALYTIC):
        out = graph.predictor_step(step, logits, x, coef)
        unmasked = x != V - 1
        assert torch.equal(out[unmasked], x[unmasked])
        assert int(((out < 0) | (out >= V)).sum()) == 0
    out = graph.predictor_step(L.STEP_DENOISE, logits, x, coef)
    assert int((out == V - 1).sum()) == 0
    assert torch.equal(out[x != V - 1], x[x != V - 1])


def test_abi_rejects_bad_arguments():
    from protein_lig_sedd_amd import ops
    from protein_lig_sedd_amd._lib import SeddError
    GL = _graphs()
    with pytest.raises((SeddError, ValueError)):
        ops.predictor_step(GL.Absorbing(25), 7, torch.zeros(1, 4, 26, device=DEV),
                           torch.zeros(1, 4, dtype=torch.int64, device=DEV), torch.zeros(1, device=DEV),
                           torch.rand(1, 4, 26, device=DEV))
    with pytest.raises(ValueError):
        ops.PhiloxNoise(1, 2 ** 32)
    with pytest.raises(RuntimeError):
        ops.gumbel_argmax(torch.rand(4, 26), torch.rand(4, 26))  # CPU tensors: no fallback
    with pytest.raises(ValueError):
        GL.get_graph(type("C", (), {"graph": type("G", (), {"type": "bogus"}), "tokens": 3}))
