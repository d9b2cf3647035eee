"""Pins the CPU oracle (oracle/sedd_cpu.py) against golden vectors captured from the imported reference
(oracle/gen_golden.py).  CPU only.  Tolerances: integer outputs bit-exact; fp32 within 1e-5 rel unless noted."""
import math

import torch

from conftest import assert_close
from oracle import sedd_cpu as O
from oracle.weights import seeded_state_dict


def test_g1_noise(golden):
    g = golden("g1_noise")
    t = g["t"]
    for name, fn in (("loglinear", O.make_noise("loglinear", eps=1e-3)),
                     ("loglinear01", O.make_noise("loglinear", eps=0.1)),
                     ("geometric", O.make_noise("geometric", sigma_min=1e-4, sigma_max=20.0)),
                     ("cosine", O.make_noise("cosine", sigma_min=1e-3, sigma_max=1.0))):
        s, ds = fn(t)
        assert_close(s, g[name + "_sigma"], 1e-6, 1e-9, name)
        assert_close(ds, g[name + "_dsigma"], 1e-6, 1e-9, name)


def test_g2_sample_transition(golden):
    g = golden("g2_sample_transition")
    sigma = g["sigma"]
    for V in (26, 2365):
        t = f"absorb{V}_"
        out = O.AbsorbingCPU(V - 1).sample_transition(g[t + "x0"], sigma, g[t + "u"])
        assert torch.equal(out, g[t + "xt"])
    out = O.UniformCPU(25).sample_transition(g["uniform25_x0"], sigma, g["uniform25_u"], g["uniform25_r"])
    assert torch.equal(out, g["uniform25_xt"])
    gr = O.AbsorbingCPU(25)
    for k, s in enumerate(g["curr_steps"].tolist()):
        out = gr.sample_transition_curriculum(g["curr_x0"], sigma, s, g["curr_u"][k])
        assert torch.equal(out, g["curr_xt"][k]), f"curriculum step {s}"


def test_g3_score_entropy(golden):
    g = golden("g3_score_entropy")
    for tag, graph in (("absorb26_", O.AbsorbingCPU(25)), ("absorb37_", O.AbsorbingCPU(36)),
                       ("absorb2365_", O.AbsorbingCPU(2364)), ("uniform25_", O.UniformCPU(25))):
        score = g[tag + "score"].clone().requires_grad_(True)
        se = graph.score_entropy(score, g[tag + "sigma"], g[tag + "x"], g[tag + "x0"])
        (se * g[tag + "w"]).sum().backward()
        assert_close(se, g[tag + "out"], 2e-5, 1e-5, tag + "fwd")
        assert_close(score.grad, g[tag + "grad"], 2e-5, 1e-6, tag + "grad")


def test_g4_graph_ops(golden):
    g = golden("g4_graph_ops")
    for tag, graph in (("absorb26_", O.AbsorbingCPU(25)), ("uniform25_", O.UniformCPU(25))):
        i, score, sigma = g[tag + "i"], g[tag + "score"], g[tag + "sigma"]
        assert_close(graph.rate(i), g[tag + "rate"], 1e-6, 0)
        assert_close(graph.transp_rate(i), g[tag + "transp_rate"], 1e-6, 0)
        assert_close(graph.reverse_rate(i, score), g[tag + "reverse_rate"], 1e-5, 1e-6)
        assert_close(graph.staggered_score(score, sigma), g[tag + "staggered_score"], 1e-5, 1e-6)
        assert_close(graph.transp_transition(i, sigma), g[tag + "transp_transition"], 1e-6, 1e-8)
        out = graph.sample_rate(i, g[tag + "sample_rate_rate"], g[tag + "sample_rate_u"])
        assert torch.equal(out, g[tag + "sample_rate_out"])
    assert torch.equal(O.AbsorbingCPU(25).sample_limit(3, 20), g["absorb26_sample_limit"])


def test_g5_sample_categorical(golden):
    g = golden("g5_sample_categorical")
    for V in (26, 37, 2365):
        assert torch.equal(O.gumbel_argmax(g[f"v{V}_p"], g[f"v{V}_u"]), g[f"v{V}_out"])


def test_g6_predictor_steps(golden):
    g = golden("g6_predictor_steps")
    noise = O.make_noise("loglinear", eps=1e-3)
    for tag, V in (("v26", 26), ("v2365", 2365)):
        graph = O.AbsorbingCPU(V - 1)
        x, logits, t, dt = g[tag + "_x"], g[tag + "_logits"], g[tag + "_t"], float(g[tag + "_dt"])
        assert torch.equal(O.euler_step(graph, logits, x, noise(t), dt, g[tag + "_euler_u"]), g[tag + "_euler_out"])
        dsig = noise(t)[0] - noise(t - dt)[0]
        assert torch.equal(O.analytic_step(graph, logits, x, dsig, g[tag + "_analytic_u"]), g[tag + "_analytic_out"])
        sig = noise(g[tag + "_denoise_t"])[0]
        assert torch.equal(O.denoise_step(graph, logits, x, sig, g[tag + "_denoise_u"]), g[tag + "_denoise_out"])
    noise = O.make_noise("geometric", sigma_min=1e-4, sigma_max=20.0)
    graph = O.UniformCPU(25)
    tag = "u25"
    x, logits, t, dt = g[tag + "_x"], g[tag + "_logits"], g[tag + "_t"], float(g[tag + "_dt"])
    assert torch.equal(O.euler_step(graph, logits, x, noise(t), dt, g[tag + "_euler_u"]), g[tag + "_euler_out"])
    dsig = noise(t)[0] - noise(t - dt)[0]
    assert torch.equal(O.analytic_step(graph, logits, x, dsig, g[tag + "_analytic_u"]), g[tag + "_analytic_out"])
    sig = noise(g[tag + "_denoise_t"])[0]
    assert torch.equal(O.denoise_step(graph, logits, x, sig, g[tag + "_denoise_u"]), g[tag + "_denoise_out"])


def test_g7_timestep_embed(golden):
    g = golden("g7_timestep_embed")
    shapes = {"mlp.0.weight": (128, 256), "mlp.0.bias": (128,), "mlp.2.weight": (128, 128), "mlp.2.bias": (128,)}
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(shapes, seed=7).items()}
    assert_close(O.timestep_embedding(g["sigma"]), g["freq"], 1e-6, 1e-7)
    c = O.sigma_cond(sd, g["sigma"], prefix="")
    assert_close(c, g["c"], 1e-5, 1e-7)
    (c * g["w"]).sum().backward()
    for k, v in sd.items():
        assert_close(v.grad, g["grad." + k], 1e-4, 1e-7, k)


def _block_shapes(D, C, cross=False):
    s = {"norm1.weight": (D,), "norm2.weight": (D,), "attn_qkv.weight": (3 * D, D), "attn_out.weight": (D, D),
         "mlp.0.weight": (4 * D, D), "mlp.0.bias": (4 * D,), "mlp.2.weight": (D, 4 * D), "mlp.2.bias": (D,),
         "adaLN_modulation.weight": (6 * D, C), "adaLN_modulation.bias": (6 * D,)}
    if cross:
        s.update({"attn_qkv.bias": (3 * D,), "attn_out.bias": (D,), "attn.in_proj_weight": (3 * D, D),
                  "attn.in_proj_bias": (3 * D,), "attn.out_proj.weight": (D, D), "attn.out_proj.bias": (D,),
                  "cross_attn.cross_attn.in_proj_weight": (3 * D, D), "cross_attn.cross_attn.in_proj_bias": (3 * D,),
                  "cross_attn.cross_attn.out_proj.weight": (D, D), "cross_attn.cross_attn.out_proj.bias": (D,),
                  "cross_attn.norm.weight": (D,), "cross_attn.norm.bias": (D,)})
    return s


def test_g8_ddit_block_flash(golden):
    g = golden("g8_ddit_block")
    for tag in ("d32", "d64", "d96"):
        t = f"flash_{tag}_"
        D, H, C, B, L = g[t + "dims"].tolist()
        sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(_block_shapes(D, C), seed=8, std=0.05).items()}
        x, c = g[t + "x"].clone().requires_grad_(True), g[t + "c"].clone().requires_grad_(True)
        y = O.ddit_block(sd, "", x, c, H)
        assert_close(y, g[t + "y"], 1e-4, 2e-5, t + "y")
        (y * g[t + "w"]).sum().backward()
        assert_close(x.grad, g[t + "dx"], 1e-4, 2e-5, t + "dx")
        assert_close(c.grad, g[t + "dc"], 1e-4, 2e-4, t + "dc")
        for k, v in sd.items():
            assert_close(v.grad, g[t + "grad." + k], 1e-4, 2e-4, t + k)


def test_g8_ddit_block_cross(golden):
    g = golden("g8_ddit_block")
    t = "cross_"
    D, H, C, B, L, Lo = g[t + "dims"].tolist()
    sd = {k: v.requires_grad_(True) for k, v in
          seeded_state_dict(_block_shapes(D, C, cross=True), seed=9, std=0.05).items()}
    x, c = g[t + "x"].clone().requires_grad_(True), g[t + "c"].clone().requires_grad_(True)
    other = g[t + "other"].clone().requires_grad_(True)
    assert_close(O.cross_ddit_block(sd, "", x, c, H), g[t + "y_noother"], 1e-4, 2e-5)
    assert_close(O.cross_ddit_block(sd, "", x, c, H, other, g[t + "mask"]), g[t + "y_masked"], 1e-4, 2e-5)
    y = O.cross_ddit_block(sd, "", x, c, H, other)
    assert_close(y, g[t + "y"], 1e-4, 2e-5)
    (y * g[t + "w"]).sum().backward()
    assert_close(x.grad, g[t + "dx"], 1e-4, 2e-5)
    assert_close(other.grad, g[t + "dother"], 1e-4, 2e-5)
    assert_close(c.grad, g[t + "dc"], 1e-4, 2e-4)
    for k, v in sd.items():
        assert_close(v.grad, g[t + "grad." + k], 1e-4, 2e-4, k)


def test_g9_cross_attention(golden):
    g = golden("g9_cross_attention")
    D, H, B, Lq, Lk = g["dims"].tolist()
    shapes = {"cross_attn.in_proj_weight": (3 * D, D), "cross_attn.in_proj_bias": (3 * D,),
              "cross_attn.out_proj.weight": (D, D), "cross_attn.out_proj.bias": (D,),
              "norm.weight": (D,), "norm.bias": (D,)}
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(shapes, seed=10, std=0.08).items()}
    x, ctx = g["x"].clone().requires_grad_(True), g["ctx"].clone().requires_grad_(True)
    y = O.cross_attention_layer(sd, "", x, ctx, H)
    assert_close(y, g["y"], 1e-5, 1e-6)
    (y * g["w"]).sum().backward()
    assert_close(x.grad, g["dx"], 1e-4, 1e-6)
    assert_close(ctx.grad, g["dctx"], 1e-4, 1e-6)
    for k, v in sd.items():
        assert_close(v.grad, g["grad." + k], 1e-4, 1e-5, k)
    x.grad = ctx.grad = None
    ym = O.cross_attention_layer(sd, "", x, ctx, H, g["mask"])
    assert_close(ym, g["y_masked"], 1e-5, 1e-6)
    (ym * g["w"]).sum().backward()
    assert_close(x.grad, g["dx_masked"], 1e-4, 1e-6)
    assert_close(ctx.grad, g["dctx_masked"], 1e-4, 1e-6)


def sedd_shapes(D, H, C, nb, V):
    s = {"vocab_embed.embedding.weight": (V, D), "sigma_map.mlp.0.weight": (C, 256), "sigma_map.mlp.0.bias": (C,),
         "sigma_map.mlp.2.weight": (C, C), "sigma_map.mlp.2.bias": (C,),
         "output_layer.norm_final.weight": (D,), "output_layer.linear.weight": (V, D),
         "output_layer.linear.bias": (V,), "output_layer.adaLN_modulation.weight": (2 * D, C),
         "output_layer.adaLN_modulation.bias": (2 * D,)}
    for i in range(nb):
        for k, v in _block_shapes(D, C).items():
            s[f"blocks.{i}.{k}"] = v
    return s


def test_g10_sedd_model(golden):
    g = golden("g10_sedd_model")
    graph, noise = O.AbsorbingCPU(25), O.make_noise("loglinear", eps=1e-3)
    for tag in ("mini", "mini96", "tiny"):
        t_ = tag + "_"
        D, H, C, nb, tokens = g[t_ + "cfg"].tolist()
        sd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=10)
        if tag != "tiny":
            sd = {k: v.requires_grad_(True) for k, v in sd.items()}
        with torch.set_grad_enabled(tag != "tiny"):
            loss, logits = O.sedd_loss(sd, graph, noise, g[t_ + "x0"], g[t_ + "t"], g[t_ + "xt"], H, nb)
        assert_close(logits, g[t_ + "logits"], 1e-4, 1e-4, tag + " logits")
        assert_close(loss, g[t_ + "loss"], 1e-5, 0, tag + " loss")
        if tag != "tiny":
            loss.backward()
            for k, v in sd.items():
                assert_close(v.grad, g[t_ + "grad." + k], 1e-3, 1e-4 * float(g[t_ + "grad." + k].abs().max()), k)
    # KAT-init-loss: with the reference's zero-init the logits are closed-form
    x0, t, xt = g["init_x0"], g["init_t"], g["init_xt"]
    sigma, dsigma = noise(t)
    kat = -O.esigm1(sigma).log()[:, None, None] - math.log(25) + torch.zeros(4, 32, 26)
    kat = kat.scatter(-1, xt[..., None], torch.zeros(4, 32, 1))
    assert_close(g["init_logits"], kat, 1e-6, 1e-6)
    se = graph.score_entropy(kat, sigma[:, None], xt, x0)
    assert_close((dsigma[:, None] * se).mean(), g["init_loss"], 1e-5, 0)
    masked = xt == 25
    assert_close(se[masked], (math.log(25) / O.esigm1(sigma)[:, None].expand_as(se))[masked], 1e-4, 1e-5)


def cross_shapes(D, C, nb):
    s = {"transformer.protein_embed.embedding.weight": (38, D), "transformer.ligand_embed.embedding.weight": (2366, D),
         "transformer.sigma_map.mlp.0.weight": (C, 256), "transformer.sigma_map.mlp.0.bias": (C,),
         "transformer.sigma_map.mlp.2.weight": (C, C), "transformer.sigma_map.mlp.2.bias": (C,)}
    for outp, V in (("protein_output", 37), ("ligand_output", 2365)):
        s.update({f"transformer.{outp}.norm_final.weight": (D,), f"transformer.{outp}.linear.weight": (V, D),
                  f"transformer.{outp}.linear.bias": (V,), f"transformer.{outp}.adaLN_modulation.weight": (2 * D, C),
                  f"transformer.{outp}.adaLN_modulation.bias": (2 * D,)})
    for tr in ("blocks_prot", "blocks_lig"):
        for i in range(nb):
            for k, v in _block_shapes(D, C, cross=True).items():
                s[f"transformer.{tr}.{i}.{k}"] = v
    return s


def test_g11_cross_model(golden):
    g = golden("g11_cross_model")
    D, H, C, nb = g["cfg"].tolist()
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(cross_shapes(D, C, nb), seed=11, std=0.05).items()}
    pt, lt, p0, l0, sigma, dsigma = g["pt"], g["lt"], g["p0"], g["l0"], g["sigma"], g["dsigma"]
    lp, ll = O.cross_joint(sd, pt, lt, sigma, H, nb)
    assert_close(lp, g["joint_prot_logits"], 1e-4, 1e-4)
    assert_close(ll, g["joint_lig_logits"], 1e-4, 1e-4)
    se_p = O.AbsorbingCPU(36).score_entropy(lp, sigma[:, None], pt, p0)
    se_l = O.AbsorbingCPU(2364).score_entropy(ll, sigma[:, None], lt, l0)
    loss_q3 = (dsigma[:, None] * ((se_p.mean() + se_l.mean()) / 2)).mean()
    assert_close(loss_q3, g["joint_loss_q3"], 1e-5, 0)
    assert_close(((dsigma[:, None] * se_p).mean() + (dsigma[:, None] * se_l).mean()) / 2, g["joint_loss_persample"], 1e-5, 0)
    loss_q3.backward()
    for k in ("transformer.blocks_prot.0.attn.in_proj_weight", "transformer.blocks_lig.1.cross_attn.cross_attn.out_proj.weight",
              "transformer.ligand_embed.embedding.weight", "transformer.sigma_map.mlp.0.weight"):
        ref = g["joint_grad." + k]
        assert_close(sd[k].grad, ref, 1e-3, 1e-4 * float(ref.abs().max()), k)
    with torch.no_grad():
        assert_close(O.cross_single_track(sd, pt, sigma, "protein", H, nb), g["prot_only_logits"], 1e-4, 1e-4)
        assert_close(O.cross_single_track(sd, lt, sigma, "ligand", H, nb), g["lig_only_logits"], 1e-4, 1e-4)
        ctx = O.cross_context(sd, p0, sigma, "protein", H, nb)
        assert_close(O.cross_single_track(sd, lt, sigma, "ligand", H, nb, other=ctx), g["lig_given_prot_logits"], 1e-4, 1e-4)
        ctx = O.cross_context(sd, l0, sigma, "ligand", H, nb)
        assert_close(O.cross_single_track(sd, pt, sigma, "protein", H, nb, other=ctx), g["prot_given_lig_logits"], 1e-4, 1e-4)


def test_g12_train_trajectory(golden):
    g = golden("g12_train_trajectory")
    D, H, C, nb, tokens = g["cfg"].tolist()
    lr, b1, b2, eps, wd, warmup, clip, decay = g["hyper"].tolist()
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=12).items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    ema = {k: v.detach().clone() for k, v in sd.items()}
    graph, noise = O.AbsorbingCPU(tokens), O.make_noise("loglinear", eps=1e-3)
    for s in range(4):
        for p in sd.values():
            p.grad = None
        loss, _ = O.sedd_loss(sd, graph, noise, g[f"x0_{s}"], g[f"t_{s}"], g[f"xt_{s}"], H, nb)
        loss.backward()
        assert_close(loss, g["losses"][s], 2e-4, 0, f"loss step {s}")
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in sd.values()))
        assert_close(gn, g["grad_norms"][s], 1e-3, 0, f"grad norm step {s}")
        coef = O.clip_coef(gn, clip)
        lr_s = lr * min((s + 1) / warmup, 1.0)
        with torch.no_grad():
            for k, p in sd.items():
                O.adamw_step(p, p.grad, m[k], v2[k], s + 1, lr_s, b1, b2, eps, wd, coef)
                O.ema_update(ema[k], p, O.ema_decay(s + 1, decay))
    for k in ("blocks.0.attn_qkv.weight", "output_layer.linear.bias", "vocab_embed.embedding.weight"):
        assert_close(sd[k], g["final." + k], 1e-3, 2e-5, k)
    assert_close(ema["blocks.0.attn_qkv.weight"], g["ema.blocks.0.attn_qkv.weight"], 1e-3, 2e-5)


def test_g13_pc_sampler(golden):
    g = golden("g13_pc_sampler")
    D, H, C, nb, tokens = g["cfg"].tolist()
    B, L, steps = g["dims"].tolist()
    sd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=13, std=0.1)
    graph, noise = O.AbsorbingCPU(tokens), O.make_noise("loglinear", eps=1e-3)
    eps = 1e-5
    ts = torch.linspace(1, eps, steps + 1)
    dt = (1 - eps) / steps
    with torch.no_grad():
        x = graph.sample_limit(B, L)
        for i in range(steps):
            t = ts[i] * torch.ones(B, 1)
            sig = noise(t)
            logits = O.sedd_forward(sd, x, sig[0].reshape(-1), H, nb)
            x = O.euler_step(graph, logits, x, sig, dt, g["euler_u"][i])
        t = ts[-1] * torch.ones(B, 1)
        sig = noise(t)[0]
        x = O.denoise_step(graph, O.sedd_forward(sd, x, sig.reshape(-1), H, nb), x, sig, g["euler_u_denoise"])
        assert torch.equal(x, g["euler_out"])
        x = graph.sample_limit(B, L)
        for i in range(steps):
            t = ts[i] * torch.ones(B, 1)
            sig = noise(t)[0]
            dsig = sig - noise(t - dt)[0]
            x = O.analytic_step(graph, O.sedd_forward(sd, x, sig.reshape(-1), H, nb), x, dsig, g["analytic_u"][i])
            assert torch.equal(x, g["analytic_steps"][i]), f"analytic step {i}"
        t = ts[-1] * torch.ones(B, 1)
        sig = noise(t)[0]
        x = O.denoise_step(graph, O.sedd_forward(sd, x, sig.reshape(-1), H, nb), x, sig, g["analytic_u_denoise"])
        assert torch.equal(x, g["analytic_out"])


def test_g14_ema(golden):
    g = golden("g14_ema")
    sw, sb = g["w0"].clone(), g["b0"].clone()
    for s in range(12):
        d = O.ema_decay(s + 1, 0.999)
        O.ema_update(sw, g["ws"][s], d)
        O.ema_update(sb, g["bs"][s], d)
    assert_close(sw, g["shadow_w"], 1e-6, 1e-7)
    assert_close(sb, g["shadow_b"], 1e-6, 1e-7)


def test_g17_bf16_autocast_pins_the_oracles_rounding_points(golden):
    """g17 = the REFERENCE under bf16 autocast (its own autocast regions run as CPU autocast, oracle/gen_golden.py).  The
    oracle's bf16 restatement (`bf16=True`: Linear inputs / weights / outputs rounded to bf16; LayerNorm, softmax in fp32;
    rotary op-by-op in bf16; the TorchScript-fused modulate / gated residual widened once and kept in fp32) reproduces it
    BIT FOR BIT — block output, every gradient, model logits — i.e. the rounding points sit where the reference's do.
    (The loss differs at bf16 level: the reference evaluates score_entropy's exp / sum on the bf16 logits in bf16, the
    oracle and the HIP kernel widen to fp32 first; bound 2e-2 per north_star.)"""
    g = golden("g17_bf16_autocast")
    D, H, C, B, L = g["block_dims"].tolist()
    assert int(g["block_y_is_fp32"]) == 1 and int(g["mini_logits_is_bf16"]) == 1  # residual fp32, logits bf16
    sd = {k: v.clone().requires_grad_(True) for k, v in seeded_state_dict(_block_shapes(D, C), seed=8, std=0.05).items()}
    x, c = g["block_x"].clone().requires_grad_(True), g["block_c"].clone().requires_grad_(True)
    y = O.ddit_block(sd, "", x, c, H, bf16=True)
    (y.float() * g["block_w"]).sum().backward()
    assert torch.equal(y.detach().float(), g["block_y"])
    assert torch.equal(x.grad, g["block_dx"]) and torch.equal(c.grad, g["block_dc"])
    for k, v in sd.items():
        assert_close(v.grad, g["block_grad." + k], 1e-6, 1e-9, k)
    with torch.no_grad():  # the fp32 path is measurably different from the bf16 golden: the fixture discriminates
        y32 = O.ddit_block({k: v.detach() for k, v in sd.items()}, "", g["block_x"], g["block_c"], H, bf16=False)
    assert float((y32 - g["block_y"]).norm() / g["block_y"].norm()) > 3e-4
    D, H, C, nb, tokens = g["mini_cfg"].tolist()
    msd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=10)
    graph, noise = O.AbsorbingCPU(tokens), O.make_noise("loglinear", eps=1e-3)
    loss, logits = O.sedd_loss(msd, graph, noise, g["mini_x0"], g["mini_t"], g["mini_xt"], H, nb, bf16=True)
    assert logits.dtype == torch.bfloat16
    assert torch.equal(logits.float(), g["mini_logits"])
    assert_close(loss, g["mini_loss"], 2e-2, 0, "loss")


def test_g18_conditional_sampling(golden):
    """The oracle replays the reference's conditional sampler runs bit-exactly: infilling through proj_fun
    (run_sample_cond.py:32-44) and the *_given_* modes of the dual-track model (sampling.py:64-76)."""
    g = golden("g18_conditional_sampling")
    noise = O.make_noise("loglinear", eps=1e-3)
    D, H, C, nb, tokens = g["infill_cfg"].tolist()
    B, L, steps = g["infill_dims"].tolist()
    sd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=13, std=0.1)
    graph = O.AbsorbingCPU(tokens)
    locs, ids = g["infill_locs"].tolist(), g["infill_ids"]
    ts = torch.linspace(1, 1e-5, steps + 1)
    dt = (1 - 1e-5) / steps
    with torch.no_grad():
        x = graph.sample_limit(B, L)
        for i in range(steps):
            x[:, locs] = ids
            sg = noise(ts[i] * torch.ones(B, 1))
            x = O.euler_step(graph, O.sedd_forward(sd, x, sg[0].reshape(-1), H, nb), x, sg, dt, g["infill_u"][i])
        x[:, locs] = ids
        sg = noise(ts[-1] * torch.ones(B, 1))[0]
        x = O.denoise_step(graph, O.sedd_forward(sd, x, sg.reshape(-1), H, nb), x, sg, g["infill_u_denoise"])
        x[:, locs] = ids
    assert torch.equal(x, g["infill_out"])
    assert torch.equal(g["infill_out"][:, locs], ids)  # the clamp holds in the reference's own output
    D, H, C, nb = g["cond_cfg"].tolist()
    B, Lp, Ll, steps = g["cond_dims"].tolist()
    csd = seeded_state_dict(cross_shapes(D, C, nb), seed=11, std=0.05)
    ts = torch.linspace(1, 1e-5, steps + 1)
    dt = (1 - 1e-5) / steps
    for tag, track, cond_track, cond, tok, Lx in (("lgp", "ligand", "protein", g["cond_prot"], 2364, Ll),
                                                  ("pgl", "protein", "ligand", g["cond_lig"], 36, Lp)):
        graph = O.AbsorbingCPU(tok)
        with torch.no_grad():
            x = graph.sample_limit(B, Lx)
            for i in range(steps + 1):
                sg = noise(ts[i] * torch.ones(B, 1))
                sig = sg[0].reshape(-1)
                ctx = O.cross_context(csd, cond, sig, cond_track, H, nb)
                logits = O.cross_single_track(csd, x, sig, track, H, nb, other=ctx)
                if i < steps:
                    x = O.euler_step(graph, logits, x, sg, dt, g[tag + "_u"][i])
                else:
                    x = O.denoise_step(graph, logits, x, sg[0], g[tag + "_u_denoise"])
        assert torch.equal(x, g[tag + "_out"]), tag


def test_g19_varlen_block(golden):
    """The oracle's variable-length attention (the flash-attn varlen hook, transformer.py:173-179) against the reference
    block run with unequal `seqlens` through the generator's stub: forward and every gradient."""
    g = golden("g19_varlen_block")
    for tag in ("a", "b"):
        t = tag + "_"
        D, H, C, B, L = g[t + "dims"].tolist()
        sd = {k: v.clone().requires_grad_(True) for k, v in seeded_state_dict(_block_shapes(D, C), seed=19, std=0.05).items()}
        x, c = g[t + "x"].clone().requires_grad_(True), g[t + "c"].clone().requires_grad_(True)
        y = O.ddit_block(sd, "", x, c, H, seqlens=g[t + "seqlens"])
        assert_close(y, g[t + "y"], 1e-5, 1e-5, tag + " y")
        (y * g[t + "w"]).sum().backward()
        assert_close(x.grad, g[t + "dx"], 1e-4, 1e-5, tag + " dx")
        assert_close(c.grad, g[t + "dc"], 1e-4, 1e-4, tag + " dc")
        for k, v in sd.items():
            assert_close(v.grad, g[t + "grad." + k], 1e-4, 1e-4, tag + " " + k)
        with torch.no_grad():  # the hook matters: one-segment-per-row attention gives a different result
            assert float((O.ddit_block({k: v.detach() for k, v in sd.items()}, "", g[t + "x"], g[t + "c"], H) - g[t + "y"]).abs().max()) > 1e-3
            assert_close(O.ddit_block({k: v.detach() for k, v in sd.items()}, "", g[t + "x"], g[t + "c"], H), g[t + "y_rows"], 1e-5, 1e-5)


def sedd_cond_state(D, H, C, nb, V, esm_dim, mol_dim, seed):
    """The g20 recipe (oracle/gen_golden.py::_load_sedd_cond): seeded weights by name, BatchNorm scales 1 + N(0, 0.02),
    running statistics from a generator seeded seed + 7 (esm_norm first)."""
    shapes = sedd_shapes(D, H, C, nb, V)
    shapes.update({"esm_proj.weight": (C, esm_dim), "esm_proj.bias": (C,), "mol_proj.weight": (C, mol_dim),
                   "mol_proj.bias": (C,), "esm_norm.weight": (C,), "esm_norm.bias": (C,), "mol_norm.weight": (C,),
                   "mol_norm.bias": (C,)})
    sd = seeded_state_dict(shapes, seed=seed)
    gen = torch.Generator().manual_seed(seed + 7)
    for nm in ("esm_norm", "mol_norm"):
        sd[nm + ".weight"] = sd[nm + ".weight"] + 1.0
    for nm in ("esm_norm", "mol_norm"):
        sd[nm + ".running_mean"] = torch.randn(C, generator=gen) * 0.05
        sd[nm + ".running_var"] = 1.0 + 0.3 * torch.rand(C, generator=gen)
    return sd


BN_BUFFERS = ("esm_norm.running_mean", "esm_norm.running_var", "mol_norm.running_mean", "mol_norm.running_var")


def test_g20_sedd_conditioned(golden):
    """`SEDD.forward(indices, sigma, esm_cond, mol_cond)` (transformer.py:277-325): pooled-conditioner projections +
    BatchNorm1d (train and eval statistics) + the mean over the given terms; and the config_large dims (V 2397, L 1024)."""
    g = golden("g20_sedd_conditioned")
    D, H, C, nb, tokens, ed, md = g["mini_cfg"].tolist()
    sd = sedd_cond_state(D, H, C, nb, tokens + 1, ed, md, seed=20)
    for k in BN_BUFFERS:
        assert_close(sd[k], g["mini_" + k + "0"], 0, 0, k)
    graph, noise = O.AbsorbingCPU(tokens), O.make_noise("loglinear", eps=1e-3)
    xt, x0, sigma = g["mini_xt"], g["mini_x0"], g["mini_sigma"]
    esm, mol = g["mini_esm"], g["mini_mol"]
    with torch.no_grad():
        assert_close(O.sedd_forward(sd, xt, sigma, H, nb, esm_cond=esm, mol_cond=mol), g["mini_eval_logits"], 1e-4, 1e-4)
        assert_close(O.sedd_forward(sd, xt, sigma, H, nb, esm_cond=esm), g["mini_esm_eval_logits"], 1e-4, 1e-4)
        assert_close(O.sedd_forward(sd, xt, sigma, H, nb), g["mini_none_eval_logits"], 1e-4, 1e-4)
    assert float((g["mini_eval_logits"] - g["mini_none_eval_logits"]).abs().max()) > 1e-2   # the conditioners matter
    sdg = {k: (v.clone().requires_grad_(True) if k not in BN_BUFFERS else v.clone()) for k, v in sd.items()}
    logits = O.sedd_forward(sdg, xt, sigma, H, nb, esm_cond=esm, mol_cond=mol, training=True)
    assert_close(logits, g["mini_train_logits"], 1e-4, 1e-4, "train logits")
    dsigma = noise(g["mini_t"])[1]
    loss = (dsigma[:, None] * graph.score_entropy(logits, sigma[:, None], xt, x0)).mean()
    assert_close(loss, g["mini_train_loss"], 1e-5, 0)
    loss.backward()
    for k, v in sdg.items():
        if k in BN_BUFFERS:
            assert_close(v, g["mini_" + k + "1"], 1e-5, 1e-6, k)
        else:
            ref = g["mini_train_grad." + k]
            # (a projection bias in front of a training-mode BatchNorm has an exactly-zero gradient: rounding noise only)
            assert_close(v.grad, ref, 1e-3, max(1e-4 * float(ref.abs().max()), 1e-7), k)
    assert int(g["mini_esm_norm.num_batches_tracked1"]) == 1
    with torch.no_grad():
        st = {k: v.detach().clone() for k, v in sdg.items()}
        assert_close(O.sedd_forward(st, xt, sigma, H, nb, esm_cond=esm, training=True), g["mini_esm_train_logits"], 1e-4, 1e-4)
        assert_close(st["esm_norm.running_mean"], g["mini_esm_norm.running_mean2"], 1e-5, 1e-6)
        assert_close(st["mol_norm.running_mean"], g["mini_mol_norm.running_mean2"], 1e-5, 1e-6)   # untouched: == ...mean1
        assert torch.equal(g["mini_mol_norm.running_mean2"], g["mini_mol_norm.running_mean1"])
    # config_large dims
    D, H, C, nb, tokens, ed, md = g["large_cfg"].tolist()
    sd = sedd_cond_state(D, H, C, nb, tokens + 1, ed, md, seed=24)
    graph = O.AbsorbingCPU(tokens)
    with torch.no_grad():
        logits = O.sedd_forward(sd, g["large_xt"], g["large_sigma"], H, nb, esm_cond=g["large_esm"], mol_cond=g["large_mol"])
        assert_close(logits[:, g["large_rows"]], g["large_logits_rows"], 1e-4, 1e-4, "large logits")
        assert_close(logits.double().sum(-1).float(), g["large_logits_rowsum"], 1e-4, 1e-2, "large row sums")
        se = graph.score_entropy(logits, g["large_sigma"][:, None], g["large_xt"], g["large_x0"])
        assert_close(se, g["large_se"], 1e-4, 1e-4, "large score entropy")
        assert_close((noise(g["large_t"])[1][:, None] * se).mean(), g["large_loss"], 1e-5, 0)


def test_philox4x32_10_known_answers():
    """The oracle's Philox4x32-10 against the Random123 known-answer vectors (kat_vectors: zero, all-ones and pi-digit
    counters / keys) — the generator sedd_predictor_step_philox runs in-kernel; and the uniform map's range."""
    import numpy as np
    kat = (([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]))
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]
        assert [int(v) for v in got] == want
    u = O.philox_uniform(1234, 7, 4096, 26)
    assert u.shape == (4096, 26) and float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 0.01
    assert not torch.equal(u, O.philox_uniform(1234, 8, 4096, 26))
    assert torch.equal(u[:, :25], O.philox_uniform(1234, 7, 4096, 25))  # slot j does not depend on the row width
