"""GPU parity of the dual-track cross-attention model (transformers_protlig_cross drop-in) against the golden vectors
captured from the reference (G8 cross block, G9 CrossAttentionLayer, G11 full ProteinLigandDiffusionModel incl. the
as-written batch-axis self-attention, quirk Q1).  fp32 compute, 1e-4 relative."""
import types

import pytest
import torch

from conftest import assert_close, rel_err
from oracle.weights import seeded_state_dict
from test_oracle_vs_golden import _block_shapes, cross_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load_sd(module, sd):
    missing, unexpected = module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("inv_freq") for k in missing), missing


def test_g8_cross_block(golden):
    from protein_lig_sedd_amd.model.transformers_protlig_cross import DDiTBlock
    from protein_lig_sedd_amd.model.rotary import Rotary
    g = golden("g8_ddit_block")
    t = "cross_"
    D, H, C, B, L, Lo = g[t + "dims"].tolist()
    blk = DDiTBlock(D, H, 0.0, C).to(DEV)
    load_sd(blk, seeded_state_dict(_block_shapes(D, C, cross=True), seed=9, std=0.05))
    rot = Rotary(D // H).to(DEV)
    x = g[t + "x"].to(DEV).requires_grad_(True)
    c = g[t + "c"].to(DEV).requires_grad_(True)
    other = g[t + "other"].to(DEV).requires_grad_(True)
    with torch.no_grad():
        assert_close(blk(x, rot(x), c), g[t + "y_noother"], 1e-4, 5e-5, "no other")
        assert_close(blk(x, rot(x), c, other, g[t + "mask"].to(DEV)), g[t + "y_masked"], 1e-4, 5e-5, "masked")
    y = blk(x, rot(x), c, other)
    assert_close(y, g[t + "y"], 1e-4, 5e-5, "y")
    (y * g[t + "w"].to(DEV)).sum().backward()
    assert_close(x.grad, g[t + "dx"], 1e-4, 5e-5, "dx")
    assert_close(other.grad, g[t + "dother"], 1e-4, 5e-5, "dother")
    assert_close(c.grad, g[t + "dc"], 1e-4, 4e-4, "dc")
    for k, p in blk.named_parameters():
        ref = g[t + "grad." + k]
        assert rel_err(p.grad, ref) < 2e-4, k
        assert_close(p.grad, ref, 1e-4, 4e-4, k)


def test_g9_cross_attention_layer(golden):
    from protein_lig_sedd_amd.model.transformers_protlig_cross import CrossAttentionLayer
    g = golden("g9_cross_attention")
    D, H, B, Lq, Lk = g["dims"].tolist()
    shapes = {"cross_attn.in_proj_weight": (3 * D, D), "cross_attn.in_proj_bias": (3 * D,),
              "cross_attn.out_proj.weight": (D, D), "cross_attn.out_proj.bias": (D,),
              "norm.weight": (D,), "norm.bias": (D,)}
    lyr = CrossAttentionLayer(D, H, 0.0).to(DEV).eval()
    load_sd(lyr, seeded_state_dict(shapes, seed=10, std=0.08))
    with torch.no_grad():
        assert_close(lyr(g["x"].to(DEV), g["ctx"].to(DEV)), g["y"], 1e-4, 1e-5)
        assert_close(lyr(g["x"].to(DEV), g["ctx"].to(DEV), context_mask=g["mask"].to(DEV)), g["y_masked"], 1e-4, 1e-5)


def _cfg(D, H, C, nb):
    ns = types.SimpleNamespace
    return ns(graph=ns(type="absorb"), model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=nb, dropout=0.0,
                                                scale_by_sigma=True))


def test_g11_cross_model(golden):
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd import graph_lib
    g = golden("g11_cross_model")
    D, H, C, nb = g["cfg"].tolist()
    model = ProteinLigandDiffusionModel(_cfg(D, C=C, H=H, nb=nb)).to(DEV)
    load_sd(model, seeded_state_dict(cross_shapes(D, C, nb), seed=11, std=0.05))
    pt, lt, p0, l0 = (g[k].to(DEV) for k in ("pt", "lt", "p0", "l0"))
    sigma, dsigma = g["sigma"].to(DEV), g["dsigma"].to(DEV)
    model.train()
    lp, ll = model(protein_indices=pt, ligand_indices=lt, timesteps=sigma, mode="joint")
    assert rel_err(lp, g["joint_prot_logits"]) < 1e-4 and rel_err(ll, g["joint_lig_logits"]) < 1e-4
    assert_close(lp, g["joint_prot_logits"], 1e-4, 2e-4)
    assert_close(ll, g["joint_lig_logits"], 1e-4, 2e-4)
    gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
    se_p = gp.score_entropy(lp, sigma[:, None], pt, p0)
    se_l = gl.score_entropy(ll, sigma[:, None], lt, l0)
    # the trainer's as-written joint loss (quirk Q3: mean first, then x dsigma; run_train_protlig_cross_att.py:289-294)
    loss_q3 = (dsigma[:, None] * ((se_p.mean() + se_l.mean()) / 2)).mean()
    assert_close(loss_q3, g["joint_loss_q3"], 1e-4, 0)
    assert_close(((dsigma[:, None] * se_p).mean() + (dsigma[:, None] * se_l).mean()) / 2, g["joint_loss_persample"], 1e-4, 0)
    loss_q3.backward()
    for k, p in model.named_parameters():
        ref = g["joint_grad." + k]
        assert_close(p.grad, ref, 1e-4, 1e-4 * float(ref.abs().max()) + 1e-7, k)
    model.eval()
    with torch.no_grad():
        assert_close(model(protein_indices=pt, timesteps=sigma, mode="protein_only"), g["prot_only_logits"], 1e-4, 2e-4)
        assert_close(model(ligand_indices=lt, timesteps=sigma, mode="ligand_only"), g["lig_only_logits"], 1e-4, 2e-4)
        assert_close(model(protein_indices=p0, ligand_indices=lt, timesteps=sigma, mode="ligand_given_protein"),
                     g["lig_given_prot_logits"], 1e-4, 2e-4)
        assert_close(model(protein_indices=pt, ligand_indices=l0, timesteps=sigma, mode="protein_given_ligand"),
                     g["prot_given_lig_logits"], 1e-4, 2e-4)
    with pytest.raises(ValueError):
        model(protein_indices=pt, timesteps=sigma, mode="nope")


def test_cross_model_sampler_and_bf16():
    """Euler pc_sampler through the dual-track model (the reference's keyword call, sampling.py:64-76) and the bf16
    opt-in vs the fp32 path (2e-2)."""
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd import graph_lib, noise_lib, sampling
    torch.manual_seed(0)
    D, H, C, nb = 64, 2, 32, 2
    sd = seeded_state_dict(cross_shapes(D, C, nb), seed=4, std=0.05)
    m32 = ProteinLigandDiffusionModel(_cfg(D, H, C, nb)).to(DEV).eval()
    m16 = ProteinLigandDiffusionModel(_cfg(D, H, C, nb), compute_dtype=torch.bfloat16).to(DEV).eval()
    load_sd(m32, sd)
    load_sd(m16, sd)
    B, Lp, Ll = 3, 24, 16
    prot = torch.randint(0, 36, (B, Lp), device=DEV)
    lig = torch.randint(0, 2364, (B, Ll), device=DEV)
    sigma = torch.rand(B, device=DEV) + 0.2
    with torch.no_grad():
        a32 = m32(protein_indices=prot, ligand_indices=lig, timesteps=sigma, mode="joint")
        a16 = m16(protein_indices=prot, ligand_indices=lig, timesteps=sigma, mode="joint")
    assert rel_err(a16[0].float(), a32[0]) < 2e-2 and rel_err(a16[1].float(), a32[1]) < 2e-2
    noise = noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    samp = sampling.get_pc_sampler(graph_lib.Absorbing(2364), noise, (B, Ll), "euler", 4, device=DEV)
    out = samp(m32, task="ligand_given_protein", prot_indices=prot)
    assert out.shape == (B, Ll) and int(((out < 0) | (out >= 2364)).sum()) == 0
    samp = sampling.get_pc_sampler(graph_lib.Absorbing(36), noise, (B, Lp), "analytic", 4, device=DEV)
    out = samp(m32, task="protein_only")
    assert out.shape == (B, Lp) and int(((out < 0) | (out >= 36)).sum()) == 0


def test_g16_cross_trainer_trajectory(golden):
    """CrossAttTrainer (joint task, quirks Q2 + Q3 as written, WarmupCosineLR) vs the 3-iteration trajectory replayed
    on the reference's model / graphs / EMA / scheduler (golden g16): losses, final weights, EMA."""
    from protein_lig_sedd_amd import graph_lib, noise_lib
    from protein_lig_sedd_amd.lr_scheduler import warmup_cosine_lr
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd.train import CrossAttTrainer
    g = golden("g16_cross_train_trajectory")
    D, H, C, nb = g["cfg"].tolist()
    b1, b2, eps, wd, clip, decay, wu, ms, blr, mlr, nlr = g["hyper"].tolist()
    model = ProteinLigandDiffusionModel(_cfg(D, H, C, nb)).to(DEV)
    load_sd(model, seeded_state_dict(cross_shapes(D, C, nb), seed=16, std=0.05))
    tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(DEV), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                         dict(lr=3e-4, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, warmup=0, grad_clip=clip),
                         task="joint", ema_decay=decay, world_size=1,
                         lr_schedule=lambda s: warmup_cosine_lr(s, int(wu), int(ms), blr, mlr, nlr))
    for it in range(3):
        batch = {"prot_tokens": g[f"p0_{it}"].to(DEV), "lig_tokens": g[f"l0_{it}"].to(DEV)}
        pert = {"prot_tokens": g[f"pt_{it}"].to(DEV), "lig_tokens": g[f"lt_{it}"].to(DEV)}
        assert tr.current_lr(tr.step_count) == pytest.approx(float(g["lrs"][it]), rel=1e-12)
        loss, lp, ll = tr.train_step(batch, "joint", t=g[f"t_{it}"].to(DEV), perturbed=pert)
        assert_close(loss, g["losses"][it], 5e-4, 0, f"loss {it}")
        assert_close(lp, g["losses_prot"][it], 5e-4, 0, f"loss_prot {it}")
        assert_close(ll, g["losses_lig"][it], 5e-4, 0, f"loss_lig {it}")
    sd = model.state_dict()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    ema = dict(zip(names, tr.ema.shadow_params))
    for key in ("transformer.blocks_prot.0.attn_qkv.weight", "transformer.blocks_lig.1.mlp.2.bias",
                "transformer.ligand_output.linear.weight", "transformer.protein_embed.embedding.weight"):
        assert_close(sd[key], g["final." + key], 2e-3, 1e-4, key)
        assert_close(ema[key], g["ema." + key], 2e-3, 1e-4, "ema " + key)
    assert tr.ema.num_updates == 3 and tr.opt_steps == 6


def test_cross_trainer_task_freezing():
    """protein_only freezes every 'lig' parameter (run_train_protlig_cross_att.py:211-215): they receive no update."""
    from protein_lig_sedd_amd import graph_lib, noise_lib
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd.train import CrossAttTrainer
    torch.manual_seed(3)
    model = ProteinLigandDiffusionModel(_cfg(64, 2, 32, 1)).to(DEV)
    load_sd(model, seeded_state_dict(cross_shapes(64, 32, 1), seed=3, std=0.05))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(DEV), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                         dict(lr=1e-3, weight_decay=0.0, warmup=0, grad_clip=1.0), task="protein_only", world_size=1)
    assert all(("lig" in n) == (not p.requires_grad) for n, p in model.named_parameters())
    loss = tr.train_step({"prot_tokens": torch.randint(0, 33, (2, 16), device=DEV)})
    assert torch.isfinite(loss)
    after = model.state_dict()
    changed = {k for k in before if not torch.equal(before[k], after[k])}
    assert changed and all("lig" not in k for k in changed)


def test_cfg3_dimensions_match_the_oracle():
    """BASELINE configs[2] at its own dimensions — D 1024, 32 heads of 32, cond 128, protein L 512 (V 37) + ligand L 128
    (V 2365), batch 4 (configs/config_prot_only.yaml:6-11, config.yaml:25, data/tokenize.py:14,25) — with one block per
    track to keep the CPU oracle quick: `joint` logits, the as-written Q3 loss and weight gradients, and `protein_only`
    logits, against oracle.cross_joint / cross_single_track (transformers_protlig_cross.py:279-320, 455-517).  The
    batch-axis MHA (quirk Q1) here is 512 (or 128) "batches" x 32 heads attending over a length-4 "sequence": a launch
    shape none of the D 64 goldens reach.  fp32 compute as the reference (autocast(float32), :401/:478): 1e-4."""
    from oracle import sedd_cpu as O
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd import graph_lib
    D, H, C, nb, B, Lp, Ll = 1024, 32, 128, 1, 4, 512, 128
    sd = seeded_state_dict(cross_shapes(D, C, nb), seed=33, std=0.03)
    model = ProteinLigandDiffusionModel(_cfg(D, H, C, nb)).to(DEV)
    load_sd(model, sd)
    gen = torch.Generator().manual_seed(33)
    p0 = torch.randint(0, 33, (B, Lp), generator=gen)
    l0 = torch.randint(0, 2362, (B, Ll), generator=gen)
    t = torch.rand(B, generator=gen) * 0.9 + 0.05
    on = O.make_noise("loglinear", eps=1e-3)
    sigma, dsigma = on(t)
    gp_cpu, gl_cpu = O.AbsorbingCPU(36), O.AbsorbingCPU(2364)
    pt = gp_cpu.sample_transition(p0, sigma[:, None], torch.rand(B, Lp, generator=gen))
    lt = gl_cpu.sample_transition(l0, sigma[:, None], torch.rand(B, Ll, generator=gen))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rp, rl = O.cross_joint(sdr, pt, lt, sigma, H, nb)
    se_p = gp_cpu.score_entropy(rp, sigma[:, None], pt, p0)
    se_l = gl_cpu.score_entropy(rl, sigma[:, None], lt, l0)
    ref_loss = (dsigma[:, None] * ((se_p.mean() + se_l.mean()) / 2)).mean()   # Q3, run_train_protlig_cross_att.py:289-294
    ref_loss.backward()
    ptd, ltd, p0d, l0d, sg, dsg = (z.to(DEV) for z in (pt, lt, p0, l0, sigma, dsigma))
    model.train()
    lp, ll = model(protein_indices=ptd, ligand_indices=ltd, timesteps=sg, mode="joint")
    assert lp.shape == (B, Lp, 37) and ll.shape == (B, Ll, 2365)
    assert rel_err(lp, rp.detach()) < 1e-4 and rel_err(ll, rl.detach()) < 1e-4
    assert_close(lp, rp.detach(), 1e-4, 2e-4, "joint protein logits")
    assert_close(ll, rl.detach(), 1e-4, 2e-4, "joint ligand logits")
    gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
    loss = (dsg[:, None] * ((gp.score_entropy(lp, sg[:, None], ptd, p0d).mean()
                             + gl.score_entropy(ll, sg[:, None], ltd, l0d).mean()) / 2)).mean()
    assert_close(loss, ref_loss.detach(), 1e-4, 0, "Q3 loss")
    loss.backward()
    named = dict(model.named_parameters())
    for k in ("transformer.blocks_prot.0.attn_qkv.weight", "transformer.blocks_prot.0.attn.in_proj_weight",
              "transformer.blocks_lig.0.cross_attn.cross_attn.in_proj_weight", "transformer.blocks_lig.0.mlp.2.weight",
              "transformer.blocks_prot.0.adaLN_modulation.weight", "transformer.ligand_output.linear.weight",
              "transformer.protein_embed.embedding.weight", "transformer.sigma_map.mlp.2.weight"):
        err = rel_err(named[k].grad, sdr[k].grad)
        assert err < 2e-4, f"{k}: relative gradient error {err:.3e}"
    model.eval()
    with torch.no_grad():
        ref_po = O.cross_single_track(sd, pt, sigma, "protein", H, nb)
        got_po = model(protein_indices=ptd, timesteps=sg, mode="protein_only")
    assert rel_err(got_po, ref_po) < 1e-4
    assert_close(got_po, ref_po, 1e-4, 2e-4, "protein_only logits")


def test_g18_conditional_sampling_bit_exact(golden):
    """SURVEY 8(f2) through the product sampler, against runs of the reference's own get_pc_sampler loop with its noise
    recorded (golden g18): (a) infilling — proj_fun clamps a prefix and a suffix before every update
    (sampling/run_sample_cond.py:32-44); (b) ligand_given_protein / protein_given_ligand on the dual-track model
    (sampling.py:64-76).  Sampled indices bit-exact for identical noise."""
    from protein_lig_sedd_amd import graph_lib, noise_lib, sampling
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from test_oracle_vs_golden import sedd_shapes
    g = golden("g18_conditional_sampling")
    noise = noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    D, H, C, nb, tokens = g["infill_cfg"].tolist()
    B, L, steps = g["infill_dims"].tolist()
    ns = types.SimpleNamespace
    model = _SEDD(ns(tokens=tokens, graph=ns(type="absorb"),
                     model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=nb, dropout=0.0, scale_by_sigma=True)),
                  compute_dtype=torch.float32).to(DEV)
    load_sd(model, seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=13, std=0.1))
    locs, ids = g["infill_locs"].tolist(), g["infill_ids"].to(DEV)

    def proj_fun(x):
        x[:, locs] = ids
        return x

    sampler = sampling.get_pc_sampler(graph_lib.Absorbing(tokens), noise, (B, L), "euler", steps, denoise=True, eps=1e-5,
                                      device=DEV, proj_fun=proj_fun)
    seq = [u.to(DEV) for u in g["infill_u"]] + [g["infill_u_denoise"].to(DEV)]
    out = proj_fun(sampler(model, noise_seq=seq))
    assert int((out.cpu() != g["infill_out"]).sum()) == 0
    assert torch.equal(out[:, locs], ids)
    # the same run with the update (clamp included) replayed as a HIP graph
    # (indexing with a Python list copies the index from the host on every call, which a capture refuses: the clamp has
    # to be device-side work)
    loc_t = torch.tensor(locs, device=DEV)

    def proj_dev(x):
        x[:, loc_t] = ids
        return x

    gsampler = sampling.get_pc_sampler(graph_lib.Absorbing(tokens), noise, (B, L), "euler", steps, denoise=True, eps=1e-5,
                                       device=DEV, proj_fun=proj_dev, hip_graph=True)
    assert int((proj_fun(gsampler(model, noise_seq=seq)).cpu() != g["infill_out"]).sum()) == 0
    bad = sampling.get_pc_sampler(graph_lib.Absorbing(tokens), noise, (B, L), "euler", steps, device=DEV, proj_fun=proj_fun,
                                  hip_graph=True)
    with pytest.raises(RuntimeError, match="could not be captured"):
        bad(model, noise_seq=seq)
    assert int((sampler(model, noise_seq=seq).cpu() != gsampler(model, noise_seq=seq).cpu()).sum()) == 0  # still usable
    # the Analytic predictor honours the same clamp (no golden: the reference's loop cannot call it, quirk Q5)
    out = proj_fun(sampling.get_pc_sampler(graph_lib.Absorbing(tokens), noise, (B, L), "analytic", steps, device=DEV,
                                           proj_fun=proj_fun)(model))
    assert torch.equal(out[:, locs], ids) and int(((out < 0) | (out >= tokens)).sum()) == 0
    # (b) conditional tracks
    D, H, C, nb = g["cond_cfg"].tolist()
    B, Lp, Ll, steps = g["cond_dims"].tolist()
    cm = ProteinLigandDiffusionModel(_cfg(D, H, C, nb)).to(DEV).eval()
    load_sd(cm, seeded_state_dict(cross_shapes(D, C, nb), seed=11, std=0.05))
    prot, lig = g["cond_prot"].to(DEV), g["cond_lig"].to(DEV)
    for tag, gr, Lx, kw in (("lgp", graph_lib.Absorbing(2364), Ll, dict(task="ligand_given_protein", prot_indices=prot)),
                            ("pgl", graph_lib.Absorbing(36), Lp, dict(task="protein_given_ligand", lig_indices=lig))):
        sampler = sampling.get_pc_sampler(gr, noise, (B, Lx), "euler", steps, denoise=True, eps=1e-5, device=DEV)
        seq = [u.to(DEV) for u in g[tag + "_u"]] + [g[tag + "_u_denoise"].to(DEV)]
        out = sampler(cm, noise_seq=seq, **kw)
        assert int((out.cpu() != g[tag + "_out"]).sum()) == 0, tag
        gsampler = sampling.get_pc_sampler(gr, noise, (B, Lx), "euler", steps, denoise=True, eps=1e-5, device=DEV,
                                           hip_graph=True)
        assert int((gsampler(cm, noise_seq=seq, **kw).cpu() != g[tag + "_out"]).sum()) == 0, tag + " (graph)"
    assert cm._ctx_step is None and cm._ctx_cache == {}  # the per-step context cache does not outlive a run


def test_context_cache_is_per_step():
    """SURVEY 8(f2) context caching: within one sampler step every call with the same conditioning indices reuses the
    conditioning track's context (its n_blocks // 2 blocks run once); the next step recomputes it (it depends on
    sigma); outside a sampler run, in training mode or under autograd nothing is cached.  Results unchanged."""
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    torch.manual_seed(1)
    D, H, C, nb, B, Lp, Ll = 64, 2, 32, 2, 2, 24, 16
    cm = ProteinLigandDiffusionModel(_cfg(D, H, C, nb)).to(DEV).eval()
    load_sd(cm, seeded_state_dict(cross_shapes(D, C, nb), seed=11, std=0.05))
    prot = torch.randint(0, 33, (B, Lp), device=DEV)
    lig = torch.randint(0, 2362, (B, Ll), device=DEV)
    s1, s2 = torch.full((B,), 0.7, device=DEV), torch.full((B,), 0.3, device=DEV)
    calls = []
    orig = cm._context_uncached
    cm._context_uncached = lambda *a: (calls.append(1), orig(*a))[1]
    with torch.no_grad():
        base1 = cm(protein_indices=prot, ligand_indices=lig, timesteps=s1, mode="ligand_given_protein")
        base2 = cm(protein_indices=prot, ligand_indices=lig, timesteps=s2, mode="ligand_given_protein")
        assert len(calls) == 2 and cm.context_cache_hits == 0          # no sampler run: never cached
        cm.sampling_step(0)
        a = cm(protein_indices=prot, ligand_indices=lig, timesteps=s1, mode="ligand_given_protein")
        b = cm(protein_indices=prot, ligand_indices=lig, timesteps=s1, mode="ligand_given_protein")
        assert len(calls) == 3 and cm.context_cache_hits == 1          # second call of the step reuses the context
        assert torch.equal(a, base1) and torch.equal(b, base1)
        # a call of the SAME step at another sigma (a corrector at its own t, a user loop) must not get the cached context
        b2 = cm(protein_indices=prot, ligand_indices=lig, timesteps=s2, mode="ligand_given_protein")
        assert len(calls) == 4 and cm.context_cache_hits == 1 and torch.equal(b2, base2)
        cm.sampling_step(1)                                            # new step = new sigma: recomputed
        c = cm(protein_indices=prot, ligand_indices=lig, timesteps=s2, mode="ligand_given_protein")
        assert len(calls) == 5 and torch.equal(c, base2)
        prot2 = prot.clone()                                           # other conditioning input: its own entry
        cm(protein_indices=prot2, ligand_indices=lig, timesteps=s2, mode="ligand_given_protein")
        assert len(calls) == 6
        cm.sampling_step(None)
        assert cm._ctx_cache == {}
    cm.sampling_step(0)
    cm.train()
    cm(protein_indices=prot, ligand_indices=lig, timesteps=s1, mode="ligand_given_protein")
    cm(protein_indices=prot, ligand_indices=lig, timesteps=s1, mode="ligand_given_protein")
    assert len(calls) == 8 and cm.context_cache_hits == 1              # training / autograd: no caching
    cm.sampling_step(None)


def test_cross_trainer_hip_graph_matches_eager(golden):
    """CrossAttTrainer(hip_graph=True): the dual-track joint step (quirks Q2 double optimiser step + Q3 weighting,
    WarmupCosineLR) replayed as one HIP graph leaves the eager step's bits — on golden g16's inputs, repeated past the
    warm-up calls and the capture."""
    from protein_lig_sedd_amd import graph_lib, noise_lib
    from protein_lig_sedd_amd.lr_scheduler import warmup_cosine_lr
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd.train import CrossAttTrainer
    g = golden("g16_cross_train_trajectory")
    D, H, C, nb = g["cfg"].tolist()
    b1, b2, eps, wd, clip, decay, wu, ms, blr, mlr, nlr = g["hyper"].tolist()

    def run(hg, side=False):
        model = ProteinLigandDiffusionModel(_cfg(D, H, C, nb), compute_dtype=torch.float32).to(DEV)
        load_sd(model, seeded_state_dict(cross_shapes(D, C, nb), seed=16, std=0.05))
        tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(DEV), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                             dict(lr=3e-4, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, warmup=0, grad_clip=clip),
                             task="joint", ema_decay=decay, world_size=1, hip_graph=hg, wgrad_stream=side,
                             lr_schedule=lambda s: warmup_cosine_lr(s, int(wu), int(ms), blr, mlr, nlr))
        out = []
        for it in range(6):
            k = it % 3
            batch = {"prot_tokens": g[f"p0_{k}"].to(DEV), "lig_tokens": g[f"l0_{k}"].to(DEV)}
            pert = {"prot_tokens": g[f"pt_{k}"].to(DEV), "lig_tokens": g[f"lt_{k}"].to(DEV)}
            out.append([float(v) for v in tr.train_step(batch, "joint", t=g[f"t_{k}"].to(DEV), perturbed=pert)])
        torch.cuda.synchronize()
        return out, tr

    oe, te = run(False)
    og, tg = run(True)
    # weight gradients on the side stream, eager and inside the captured graph (a forked capture): the same bits
    for hg in (False, True):
        os_, ts_ = run(hg, side=True)
        assert os_ == oe and torch.equal(ts_.flat.p, te.flat.p) and torch.equal(ts_.flat.ema, te.flat.ema), hg
    assert og == oe
    assert tg.opt_steps == te.opt_steps == 12 and tg.ema.num_updates == 6
    for name in ("p", "m", "v", "ema"):
        assert torch.equal(getattr(tg.flat, name), getattr(te.flat, name)), name
    assert_close(torch.tensor(oe[0][0]), g["losses"][0], 5e-4, 0)


@pytest.mark.parametrize("cdt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_cfg3_full_depth_joint_forward_matches_the_oracle(cdt, tol):
    """BASELINE configs[2] as `run_train_crossatt.sh` trains it — D 1024, 32 heads, cond 128, 12 + 12 blocks, Lp 512 + Ll 128,
    batch 4 (528 M parameters) — `joint` logits and the as-written Q3 loss against oracle.cross_joint at FULL depth (forward
    only, to keep the CPU oracle at a few seconds): fp32 as the reference computes (1e-4), and the bf16 compute path (2e-2)."""
    from oracle import sedd_cpu as O
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd import graph_lib
    D, H, C, nb, B, Lp, Ll = 1024, 32, 128, 12, 4, 512, 128
    sd = seeded_state_dict(cross_shapes(D, C, nb), seed=35, std=0.02)
    model = ProteinLigandDiffusionModel(_cfg(D, H, C, nb), compute_dtype=cdt).to(DEV).eval()
    load_sd(model, sd)
    gen = torch.Generator().manual_seed(35)
    p0 = torch.randint(0, 33, (B, Lp), generator=gen)
    l0 = torch.randint(0, 2362, (B, Ll), generator=gen)
    t = torch.rand(B, generator=gen) * 0.9 + 0.05
    sigma, dsigma = O.make_noise("loglinear", eps=1e-3)(t)
    gp_cpu, gl_cpu = O.AbsorbingCPU(36), O.AbsorbingCPU(2364)
    pt = gp_cpu.sample_transition(p0, sigma[:, None], torch.rand(B, Lp, generator=gen))
    lt = gl_cpu.sample_transition(l0, sigma[:, None], torch.rand(B, Ll, generator=gen))
    with torch.no_grad():
        rp, rl = O.cross_joint(sd, pt, lt, sigma, H, nb)
        ref_loss = (dsigma[:, None] * ((gp_cpu.score_entropy(rp, sigma[:, None], pt, p0).mean()
                                        + gl_cpu.score_entropy(rl, sigma[:, None], lt, l0).mean()) / 2)).mean()
        ptd, ltd, p0d, l0d, sg, dsg = (z.to(DEV) for z in (pt, lt, p0, l0, sigma, dsigma))
        lp, ll = model(protein_indices=ptd, ligand_indices=ltd, timesteps=sg, mode="joint")
        assert rel_err(lp.float(), rp) < tol and rel_err(ll.float(), rl) < tol
        gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
        loss = (dsg[:, None] * ((gp.score_entropy(lp, sg[:, None], ptd, p0d).mean()
                                 + gl.score_entropy(ll, sg[:, None], ltd, l0d).mean()) / 2)).mean()
    assert_close(loss, ref_loss, tol, 0, "Q3 loss")
