"""GPU parity of the drop-in modules (TimestepEmbedder, DDiTBlock, _SEDD, loss_fn, pc_sampler) against the golden
vectors captured from the reference (tests/golden, oracle/gen_golden.py) and the CPU oracle.
fp32 compute: logits / loss 1e-4 relative (north_star); bf16 compute: 2e-2; sampled indices bit-exact."""
import types

import pytest
import torch

from conftest import assert_close, rel_err
from oracle import sedd_cpu as O
from oracle.weights import seeded_state_dict
from test_oracle_vs_golden import BN_BUFFERS, _block_shapes, sedd_cond_state, sedd_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ns(**kw):
    return types.SimpleNamespace(**kw)


def sedd_config(D, H, C, nb, tokens, dropout=0.0, graph="absorb"):
    return ns(tokens=tokens, graph=ns(type=graph),
              model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=nb, dropout=dropout, scale_by_sigma=True))


def load_sd(module, sd):
    missing, unexpected = module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("inv_freq") for k in missing), missing


def test_g7_timestep_embedder(golden):
    from protein_lig_sedd_amd.model.transformer import TimestepEmbedder
    g = golden("g7_timestep_embed")
    shapes = {"mlp.0.weight": (128, 256), "mlp.0.bias": (128,), "mlp.2.weight": (128, 128), "mlp.2.bias": (128,)}
    m = TimestepEmbedder(128).to(DEV)
    load_sd(m, seeded_state_dict(shapes, seed=7))
    sigma = g["sigma"].to(DEV)
    assert_close(TimestepEmbedder.timestep_embedding(sigma, 256), g["freq"], 1e-5, 2e-6)
    c = m(sigma, silu=True)
    assert_close(c, g["c"], 1e-4, 1e-6)
    (c * g["w"].to(DEV)).sum().backward()
    for k, p in m.named_parameters():
        assert_close(p.grad, g["grad." + k], 1e-4, 1e-6, k)
    # reference call form: F.silu(sigma_map(sigma))
    assert_close(torch.nn.functional.silu(m(sigma)), g["c"], 1e-4, 1e-6)


@pytest.mark.parametrize("tag", ["d32", "d64", "d96"])
def test_g8_ddit_block_fp32(golden, tag):
    from protein_lig_sedd_amd.model.transformer import DDiTBlock
    from protein_lig_sedd_amd.model.rotary import Rotary
    g = golden("g8_ddit_block")
    t = f"flash_{tag}_"
    D, H, C, B, L = g[t + "dims"].tolist()
    blk = DDiTBlock(D, H, C, dropout=0.0, compute_dtype=torch.float32).to(DEV)
    load_sd(blk, seeded_state_dict(_block_shapes(D, C), seed=8, std=0.05))
    x = g[t + "x"].to(DEV).requires_grad_(True)
    c = g[t + "c"].to(DEV).requires_grad_(True)
    rot = Rotary(D // H).to(DEV)
    y = blk(x, rot(x), c)
    assert rel_err(y, g[t + "y"]) < 1e-5
    assert_close(y, g[t + "y"], 1e-4, 5e-5, "y")
    (y * g[t + "w"].to(DEV)).sum().backward()
    assert_close(x.grad, g[t + "dx"], 1e-4, 5e-5, "dx")
    assert_close(c.grad, g[t + "dc"], 1e-4, 4e-4, "dc")
    for k, p in blk.named_parameters():
        ref = g[t + "grad." + k]
        assert rel_err(p.grad, ref) < 1e-4, k
        assert_close(p.grad, ref, 1e-4, 4e-4, k)


def test_ddit_block_bf16_and_dropout():
    """bf16 compute path vs the oracle's bf16 restatement (2e-2), and dropout-mode sanity (mask actually applied,
    backward consistent with the saved mask)."""
    from protein_lig_sedd_amd.model.transformer import DDiTBlock
    from protein_lig_sedd_amd.model.rotary import Rotary
    torch.manual_seed(0)
    D, H, C, B, L = 256, 4, 128, 3, 96
    sd = seeded_state_dict(_block_shapes(D, C), seed=21, std=0.05)
    blk = DDiTBlock(D, H, C, dropout=0.0).to(DEV)
    load_sd(blk, sd)
    x, c = torch.randn(B, L, D), torch.randn(B, C) * 0.5
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, cr = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
    ref = O.ddit_block(sdr, "", xr, cr, H, bf16=True)
    w = torch.randn(B, L, D)
    (ref * w).sum().backward()
    xd, cd = x.to(DEV).requires_grad_(True), c.to(DEV).requires_grad_(True)
    rot = Rotary(D // H).to(DEV)
    y = blk(xd, rot(xd), cd)
    assert y.dtype == torch.float32
    assert rel_err(y, ref) < 2e-2
    (y * w.to(DEV)).sum().backward()
    assert rel_err(xd.grad, xr.grad) < 2e-2
    assert rel_err(cd.grad, cr.grad) < 3e-2
    for k, p in blk.named_parameters():
        assert rel_err(p.grad, sdr[k].grad) < 3e-2, k
    # dropout: train mode differs from eval mode, eval is deterministic
    blk2 = DDiTBlock(D, H, C, dropout=0.5, compute_dtype=torch.float32).to(DEV)
    load_sd(blk2, sd)
    blk2.train()
    y1 = blk2(xd.detach(), rot(xd), cd.detach())
    blk2.eval()
    y2 = blk2(xd.detach(), rot(xd), cd.detach())
    y3 = blk2(xd.detach(), rot(xd), cd.detach())
    assert torch.equal(y2, y3) and not torch.equal(y1, y2)
    blk2.train()
    xg = xd.detach().clone().requires_grad_(True)
    blk2(xg, rot(xd), cd.detach()).sum().backward()
    assert torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("tag", ["mini", "mini96", "tiny"])
def test_g10_sedd_model_fp32(golden, tag):
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd import graph_lib, noise_lib, losses
    g = golden("g10_sedd_model")
    t_ = tag + "_"
    D, H, C, nb, tokens = g[t_ + "cfg"].tolist()
    model = _SEDD(sedd_config(D, H, C, nb, tokens), compute_dtype=torch.float32).to(DEV)
    load_sd(model, seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=10))
    graph, noise = graph_lib.Absorbing(tokens), noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    x0, t, xt = g[t_ + "x0"].to(DEV), g[t_ + "t"].to(DEV), g[t_ + "xt"].to(DEV)
    model.train()
    sigma = noise(t)[0]
    logits = model(xt, sigma)
    assert logits.dtype == torch.float32
    assert rel_err(logits, g[t_ + "logits"]) < 1e-4
    assert_close(logits, g[t_ + "logits"], 1e-4, 2e-4, tag + " logits")
    for fused in (True, False):
        model.zero_grad(set_to_none=True)
        loss = losses.get_loss_fn(noise, graph, train=True, fused=fused)(model, x0, t=t, perturbed_batch=xt)
        assert_close(loss, g[t_ + "loss"], 1e-4, 0, tag + " loss")
        if tag != "tiny":
            loss.backward()
            for k, p in model.named_parameters():
                ref = g[t_ + "grad." + k]
                assert_close(p.grad, ref, 1e-4, 1e-4 * float(ref.abs().max()), f"{k} fused={fused}")


def test_sedd_model_bf16_vs_oracle():
    from protein_lig_sedd_amd.model.transformer import _SEDD
    torch.manual_seed(3)
    D, H, C, nb, tokens, B, L = 128, 2, 128, 2, 25, 4, 64
    sd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=33)
    model = _SEDD(sedd_config(D, H, C, nb, tokens)).to(DEV).eval()
    load_sd(model, sd)
    xt = torch.randint(0, tokens + 1, (B, L))
    sigma = torch.rand(B) * 3 + 0.1
    ref = O.sedd_forward(sd, xt, sigma, H, nb, bf16=True)
    with torch.no_grad():
        out = model(xt.to(DEV), sigma.to(DEV))
    assert out.dtype == torch.bfloat16
    assert rel_err(out.float(), ref.float()) < 2e-2
    ref32 = O.sedd_forward(sd, xt, sigma, H, nb, bf16=False)
    assert rel_err(out.float(), ref32) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("sinks", [False, True])
def test_block_boundary_fusion_matches_stand_alone_blocks(dtype, sinks):
    """_SEDD chains consecutive blocks (layers.BlockLink): block i's last gated residual + block i+1's first LayerNorm in
    one pass, their backwards in one pass.  Against the same model with the chain switched off: the forward is the same
    bits (dropout on), every gradient agrees to fp32 fold-order noise; with and without gradient sinks on the
    parameters; eval-mode (sampling) forward bit-identical too."""
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.model import layers as Lyr
    from protein_lig_sedd_amd import ops
    D, H, C, nb, tokens, B, L = 128, 2, 64, 3, 25, 3, 48
    sd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=51, std=0.05)
    torch.manual_seed(5)
    xt = torch.randint(0, tokens + 1, (B, L), device=DEV)
    sigma = torch.rand(B, device=DEV) * 3 + 0.1
    wgt = torch.randn(B, L, tokens + 1, device=DEV)

    def run(chain, train=True):
        cfg = sedd_config(D, H, C, nb, tokens)
        cfg.model.dropout = 0.1
        model = _SEDD(cfg, compute_dtype=dtype).to(DEV)
        load_sd(model, sd)
        model.train(train)
        if sinks:
            for p in model.parameters():
                p.grad = torch.zeros_like(p)
        Lyr.BLOCK_CHAIN = chain
        ops.DropSeed._counter = 0
        try:
            if not train:
                with torch.no_grad():
                    return model(xt, sigma), None
            out = model(xt, sigma)
            (out.float() * wgt).sum().backward()
        finally:
            Lyr.BLOCK_CHAIN = True
        return out.detach(), {k: p.grad.clone() for k, p in model.named_parameters()}

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert torch.equal(o1, o0)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 2e-5, k
    assert torch.equal(g1["blocks.1.attn_qkv.weight"], g0["blocks.1.attn_qkv.weight"])  # (the data path is the same bits)
    assert torch.equal(run(True, train=False)[0], run(False, train=False)[0])


def test_block_boundary_fusion_guards():
    """a forward hook that REPLACES a block's output breaks the chain silently (the next block runs stand-alone, same
    results as the unchained model); a second consumer of a block's output is refused in backward instead of producing a
    wrong gradient."""
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.model import layers as Lyr
    D, H, C, nb, tokens, B, L = 64, 2, 32, 3, 25, 2, 16
    sd = seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=52, std=0.05)
    torch.manual_seed(6)
    xt = torch.randint(0, tokens + 1, (B, L), device=DEV)
    sigma = torch.rand(B, device=DEV) * 3 + 0.1

    def build():
        model = _SEDD(sedd_config(D, H, C, nb, tokens), compute_dtype=torch.float32).to(DEV).train()
        load_sd(model, sd)
        return model

    def grads(model, extra=None):
        out = model(xt, sigma)
        loss = out.float().square().sum() + (0.0 if extra is None else extra())
        loss.backward()
        return out.detach(), {k: p.grad.clone() for k, p in model.named_parameters()}

    Lyr.BLOCK_CHAIN = False
    try:
        m0 = build()
        m0.blocks[1].register_forward_hook(lambda mod, inp, out: out * 1.0)
        o0, g0 = grads(m0)
    finally:
        Lyr.BLOCK_CHAIN = True
    m1 = build()
    m1.blocks[1].register_forward_hook(lambda mod, inp, out: out * 1.0)  # a new tensor: block 2 must not take the hand-over
    o1, g1 = grads(m1)
    assert torch.equal(o1, o0)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 2e-5, k
    m2 = build()
    seen = []
    m2.blocks[0].register_forward_hook(lambda mod, inp, out: seen.append(out))
    with pytest.raises(RuntimeError, match="second consumer"):
        grads(m2, extra=lambda: seen[0].sum())


def test_kat_init_loss():
    """SURVEY section 4 KAT: with the reference's zero-init the logits are -log(e^sigma-1) - log(V-1) and the loss
    per masked token log(V-1)/(e^sigma-1)."""
    import math
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd import graph_lib
    torch.manual_seed(0)
    model = _SEDD(sedd_config(64, 2, 128, 2, 25), compute_dtype=torch.float32).to(DEV)
    B, L, V = 4, 32, 26
    xt = torch.randint(0, V, (B, L), device=DEV)
    sigma = torch.rand(B, device=DEV) * 2 + 0.05
    logits = model(xt, sigma)
    kat = (-O.esigm1(sigma.cpu()).log()[:, None, None] - math.log(25) + torch.zeros(B, L, V))
    kat = kat.scatter(-1, xt.cpu()[..., None], torch.zeros(B, L, 1))
    assert_close(logits, kat, 1e-5, 1e-5)
    se = graph_lib.Absorbing(25).score_entropy(logits.detach(), sigma[:, None], xt, torch.zeros_like(xt))
    masked = (xt == 25).cpu()
    assert_close(se.cpu()[masked], (math.log(25) / O.esigm1(sigma.cpu())[:, None].expand(B, L))[masked], 1e-4, 1e-5)


def test_g13_pc_sampler_bit_exact(golden):
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd import graph_lib, noise_lib, sampling
    g = golden("g13_pc_sampler")
    D, H, C, nb, tokens = g["cfg"].tolist()
    B, L, steps = g["dims"].tolist()
    model = _SEDD(sedd_config(D, H, C, nb, tokens), compute_dtype=torch.float32).to(DEV)
    load_sd(model, seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=13, std=0.1))
    graph, noise = graph_lib.Absorbing(tokens), noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    for pred in ("euler", "analytic"):
        sampler = sampling.get_pc_sampler(graph, noise, (B, L), pred, steps, denoise=True, eps=1e-5, device=DEV)
        seq = [u.to(DEV) for u in g[pred + "_u"]] + [g[pred + "_u_denoise"].to(DEV)]
        out = sampler(model, noise_seq=seq)
        assert int((out.cpu() != g[pred + "_out"]).sum()) == 0, pred
        # the same run with the predictor update replayed as a HIP graph (twice: the second call reuses the capture)
        gs = sampling.get_pc_sampler(graph, noise, (B, L), pred, steps, denoise=True, eps=1e-5, device=DEV, hip_graph=True)
        for _ in range(2):
            assert int((gs(model, noise_seq=seq).cpu() != g[pred + "_out"]).sum()) == 0, pred + " (graph)"
    # without supplied noise: valid ids, no mask left after the denoiser
    out = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, device=DEV)(model)
    assert int(((out < 0) | (out >= tokens)).sum()) == 0
    # ... and under graph replay torch's generator advances per replay: valid ids, and two runs differ
    gs = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, device=DEV, hip_graph=True)
    o1, o2 = gs(model), gs(model)
    assert int(((o1 < 0) | (o1 >= tokens)).sum()) == 0 and int(((o2 < 0) | (o2 >= tokens)).sum()) == 0
    assert not torch.equal(o1, o2)
    with pytest.raises(ValueError):
        from protein_lig_sedd_amd import ops
        gs(model, noise_seq=ops.PhiloxStream(1))
    # a bf16 model whose weights are replaced after a capture: the next run must see the new weights (new capture)
    mb = _SEDD(sedd_config(D, H, C, nb, tokens), compute_dtype=torch.bfloat16).to(DEV)
    load_sd(mb, seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=13, std=0.1))
    seq = [u.to(DEV) for u in g["euler_u"]] + [g["euler_u_denoise"].to(DEV)]
    eager = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, device=DEV)
    gs = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, device=DEV, hip_graph=True)
    assert torch.equal(gs(mb, noise_seq=seq), eager(mb, noise_seq=seq))
    load_sd(mb, seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=14, std=0.1))
    assert torch.equal(gs(mb, noise_seq=seq), eager(mb, noise_seq=seq))
    # upstream positional AnalyticPredictor signature (quirk Q5) through the Graph-method kernels
    from protein_lig_sedd_amd.model.utils import get_score_fn
    pr = sampling.get_predictor("analytic")(graph, noise)
    x = graph.sample_limit(B, L).to(DEV)
    t = torch.ones(B, 1, device=DEV)
    u = g["analytic_u"][0].to(DEV)
    with torch.no_grad():
        a = pr.update_fn(get_score_fn(model, sampling=True), x, t, (1 - 1e-5) / steps, u=u)
    assert int((a.cpu() != g["analytic_steps"][0]).sum()) == 0
    with pytest.raises(KeyError):
        sampling.get_predictor("bogus")


def test_g17_bf16_path_vs_the_reference_under_autocast(golden):
    """The HIP bf16 compute path against the REFERENCE ITSELF under bf16 autocast (golden g17: the reference's autocast
    region run with CPU autocast, oracle/gen_golden.py) — block output and gradients, model logits, loss and gradients,
    at north_star's bf16 bound 2e-2.  The kernels round less often than the reference (rotary and GELU are applied to
    fp32 accumulators in GEMM epilogues), so agreement is to bf16 accuracy, not bitwise."""
    from protein_lig_sedd_amd.model.transformer import DDiTBlock, _SEDD
    from protein_lig_sedd_amd.model.rotary import Rotary
    from protein_lig_sedd_amd import graph_lib, noise_lib, losses
    g = golden("g17_bf16_autocast")
    D, H, C, B, L = g["block_dims"].tolist()
    blk = DDiTBlock(D, H, C, dropout=0.0, compute_dtype=torch.bfloat16).to(DEV)
    load_sd(blk, seeded_state_dict(_block_shapes(D, C), seed=8, std=0.05))
    x = g["block_x"].to(DEV).requires_grad_(True)
    c = g["block_c"].to(DEV).requires_grad_(True)
    rot = Rotary(D // H).to(DEV)
    y = blk(x, rot(x), c)
    assert y.dtype == torch.float32
    # the block output is residual + update: judge the update (y - x), the part the bf16 arithmetic produces
    assert rel_err(y - x.detach(), g["block_y"] - g["block_x"]) < 2e-2
    (y * g["block_w"].to(DEV)).sum().backward()
    assert rel_err(x.grad, g["block_dx"]) < 2e-2 and rel_err(c.grad, g["block_dc"]) < 2e-2
    for k, p in blk.named_parameters():
        assert rel_err(p.grad, g["block_grad." + k]) < 2e-2, k
    D, H, C, nb, tokens = g["mini_cfg"].tolist()
    model = _SEDD(sedd_config(D, H, C, nb, tokens), compute_dtype=torch.bfloat16).to(DEV)
    load_sd(model, seeded_state_dict(sedd_shapes(D, H, C, nb, tokens + 1), seed=10))
    graph, noise = graph_lib.Absorbing(tokens), noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    x0, t, xt = g["mini_x0"].to(DEV), g["mini_t"].to(DEV), g["mini_xt"].to(DEV)
    model.train()
    logits = model(xt, noise(t)[0])
    assert logits.dtype == torch.bfloat16
    assert rel_err(logits.float(), g["mini_logits"]) < 2e-2
    loss = losses.get_loss_fn(noise, graph, train=True)(model, x0, t=t, perturbed_batch=xt)
    assert_close(loss, g["mini_loss"], 2e-2, 0, "loss")
    loss.backward()
    for k, p in model.named_parameters():
        ref = g["mini_grad." + k]
        assert rel_err(p.grad, ref) < 3e-2, f"{k}: {rel_err(p.grad, ref):.3e}"


def _varlen_ref(q, k, v, cu):
    out = torch.zeros_like(q, dtype=torch.float32)
    d = q.shape[-1]
    for a, b in zip(cu[:-1], cu[1:]):
        if b > a:
            sc = torch.einsum("lhd,mhd->hlm", q[a:b].float(), k[a:b].float()) * (d ** -0.5)
            out[a:b] = torch.einsum("hlm,mhd->lhd", torch.softmax(sc, -1), v[a:b].float())
    return out


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("d", [32, 64, 96])
def test_attn_varlen_kernels(dtype, tol, d):
    """sedd_attn_varlen_fwd / _bwd on a packed qkv [B, L, 3, H, d] against exact per-segment softmax attention (fp32
    autograd): ragged segments that cross the 64-key tiles and the 128-row blocks, a 1-token and an EMPTY segment, a
    segment as long as the stated maximum, and a tail of tokens that belongs to no segment (must stay zero)."""
    from protein_lig_sedd_amd import ops
    torch.manual_seed(d)
    B, L, H = 3, 160, 2
    T = B * L
    lens = [0, 160, 1, 0, 131, 64, 97]            # sums to 453 < T = 480: 27 tokens outside every segment
    cu_list = torch.tensor(lens).cumsum(0).tolist()
    cu = torch.tensor(cu_list, dtype=torch.int32, device=DEV)
    qkv32 = (torch.randn(B, L, 3, H, d) * 0.7)
    qkv = qkv32.to(DEV).to(dtype)
    ref_in = qkv.float().cpu().reshape(T, 3, H, d).requires_grad_(True)
    ref = _varlen_ref(ref_in[:, 0], ref_in[:, 1], ref_in[:, 2], cu_list)
    w = torch.randn(T, H, d)
    (ref * w).sum().backward()
    o, lse = ops.attn_varlen_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], cu, L)
    assert o.shape == (B, L, H, d) and lse.shape == (H, T)
    assert rel_err(o.float().reshape(T, H, d), ref.detach()) < tol
    assert float(o.reshape(T, H, d)[cu_list[-1]:].abs().max()) == 0.0
    dqkv = torch.zeros_like(qkv)
    ops.attn_varlen_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, w.to(DEV).to(dtype).view(B, L, H, d), lse,
                        dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], cu, L)
    got = dqkv.float().cpu().reshape(T, 3, H, d)
    for i, nm in enumerate(("dq", "dk", "dv")):
        assert rel_err(got[:, i], ref_in.grad[:, i]) < (tol if dtype == torch.float32 else 3e-2), nm
    assert float(got[cu_list[-1]:].abs().max()) == 0.0
    # equal-length segments = the fixed-length kernel
    cu_eq = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device=DEV)
    o2, _ = ops.attn_varlen_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], cu_eq, L)
    o3, _ = ops.attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
    assert rel_err(o2.float(), o3.float()) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g19_ddit_block_varlen(golden, tag):
    """DDiTBlock.forward(seqlens=...) — the flash-attn varlen hook (transformer.py:173-179), SURVEY 8(f4) — against the
    reference block run with unequal segment lengths (golden g19): fp32 forward + all gradients at 1e-4, and the bf16
    path against the oracle at 2e-2."""
    from protein_lig_sedd_amd.model.transformer import DDiTBlock
    from protein_lig_sedd_amd.model.rotary import Rotary
    g = golden("g19_varlen_block")
    t = tag + "_"
    D, H, C, B, L = g[t + "dims"].tolist()
    sd = seeded_state_dict(_block_shapes(D, C), seed=19, std=0.05)
    blk = DDiTBlock(D, H, C, dropout=0.0, compute_dtype=torch.float32).to(DEV)
    load_sd(blk, sd)
    x = g[t + "x"].to(DEV).requires_grad_(True)
    c = g[t + "c"].to(DEV).requires_grad_(True)
    seqlens = g[t + "seqlens"].to(DEV)
    rot = Rotary(D // H).to(DEV)
    y = blk(x, rot(x), c, seqlens=seqlens)
    assert rel_err(y, g[t + "y"]) < 1e-5
    assert_close(y, g[t + "y"], 1e-4, 5e-5, "y")
    (y * g[t + "w"].to(DEV)).sum().backward()
    assert_close(x.grad, g[t + "dx"], 1e-4, 5e-5, "dx")
    assert_close(c.grad, g[t + "dc"], 1e-4, 4e-4, "dc")
    for k, p in blk.named_parameters():
        ref = g[t + "grad." + k]
        assert rel_err(p.grad, ref) < 1e-4, k
    with torch.no_grad():
        assert_close(blk(x, rot(x), c), g[t + "y_rows"], 1e-4, 5e-5, "seqlens=None")
    blk16 = DDiTBlock(D, H, C, dropout=0.0, compute_dtype=torch.bfloat16).to(DEV)
    load_sd(blk16, sd)
    with torch.no_grad():
        y16 = blk16(x, rot(x), c, seqlens=seqlens)
        ref16 = O.ddit_block(sd, "", g[t + "x"], g[t + "c"], H, bf16=True, seqlens=g[t + "seqlens"])
    assert rel_err(y16 - x.detach(), ref16.float() - g[t + "x"]) < 2e-2


def _sedd_cond_model(cfgv, seed, dtype=torch.float32):
    from protein_lig_sedd_amd.model.transformer import SEDD
    D, H, C, nb, tokens, ed, md = cfgv
    cfg = sedd_config(D, H, C, nb, tokens)
    cfg.model.esm_dim, cfg.model.molformer_dim = ed, md
    model = SEDD(cfg, compute_dtype=dtype).to(DEV)
    sd = sedd_cond_state(D, H, C, nb, tokens + 1, ed, md, seed=seed)
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("inv_freq") or k.endswith("num_batches_tracked") for k in missing), missing
    return model, sd


def test_g20_sedd_with_pooled_conditioners(golden):
    """`SEDD.forward(indices, sigma, esm_cond, mol_cond)` (transformer.py:277-325) against the reference: eval and
    training BatchNorm, the two- and three-term mean, running-statistic updates, loss and every gradient."""
    from protein_lig_sedd_amd import graph_lib, noise_lib, losses
    g = golden("g20_sedd_conditioned")
    model, _ = _sedd_cond_model(g["mini_cfg"].tolist(), seed=20)
    tokens = int(g["mini_cfg"][4])
    graph, noise = graph_lib.Absorbing(tokens), noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    xt, x0, t = g["mini_xt"].to(DEV), g["mini_x0"].to(DEV), g["mini_t"].to(DEV)
    sigma = noise(t)[0]
    esm, mol = g["mini_esm"].to(DEV), g["mini_mol"].to(DEV)
    model.eval()
    with torch.no_grad():
        for key, args in (("mini_eval_logits", (esm, mol)), ("mini_esm_eval_logits", (esm, None)),
                          ("mini_none_eval_logits", (None, None))):
            out = model(xt, sigma, *args)
            assert rel_err(out, g[key]) < 1e-4, key
            assert_close(out, g[key], 1e-4, 2e-4, key)
    model.train()
    loss = losses.get_loss_fn(noise, graph, train=True, conditioners=True)(model, x0, t=t, perturbed_batch=xt,
                                                                           esm_cond=esm, mol_cond=mol)
    assert_close(loss, g["mini_train_loss"], 1e-4, 0, "loss")
    loss.backward()
    for k, p in model.named_parameters():
        ref = g["mini_train_grad." + k]
        assert_close(p.grad, ref, 1e-4, max(1e-4 * float(ref.abs().max()), 1e-7), k)
    for k in BN_BUFFERS:
        mod, buf = k.split(".")
        assert_close(getattr(getattr(model, mod), buf), g["mini_" + k + "1"], 1e-5, 1e-6, k)
    assert int(model.esm_norm.num_batches_tracked) == 1 and int(model.mol_norm.num_batches_tracked) == 1
    with torch.no_grad():
        out = model(xt, sigma, esm, None)
    assert_close(out, g["mini_esm_train_logits"], 1e-4, 2e-4, "esm-only training forward")
    assert_close(model.esm_norm.running_mean, g["mini_esm_norm.running_mean2"], 1e-5, 1e-6)
    assert_close(model.mol_norm.running_mean, g["mini_mol_norm.running_mean2"], 1e-5, 1e-6)
    # as written, the reference's loss_fn DROPS the conditioners (losses.py:26): the default does the same
    model.eval()
    with torch.no_grad():
        l0 = losses.get_loss_fn(noise, graph, train=False)(model, x0, t=t, perturbed_batch=xt, esm_cond=esm, mol_cond=mol)
        l1 = losses.get_loss_fn(noise, graph, train=False)(model, x0, t=t, perturbed_batch=xt)
    assert torch.equal(l0, l1)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        model.train()
        model(xt[:1], sigma[:1], esm[:1], mol[:1])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_g20_config_large_dims(golden, dtype, tol):
    """BASELINE configs[3] as configs/config_large.yaml writes it — D 1024, 16 heads, cond 128, tokens 2396 (V 2397),
    L 1024, pooled conditioners 640 / 768 — against the reference (2 blocks): logits (row subsample + all row sums) and
    the score-entropy loss over the full [2, 1024, 2397] log-score."""
    from protein_lig_sedd_amd import graph_lib, noise_lib
    g = golden("g20_sedd_conditioned")
    model, _ = _sedd_cond_model(g["large_cfg"].tolist(), seed=24, dtype=dtype)
    model.eval()
    tokens = int(g["large_cfg"][4])
    graph, noise = graph_lib.Absorbing(tokens), noise_lib.LogLinearNoise(eps=1e-3).to(DEV)
    xt, x0, t = g["large_xt"].to(DEV), g["large_x0"].to(DEV), g["large_t"].to(DEV)
    sigma, dsigma = noise(t)
    with torch.no_grad():
        logits = model(xt, sigma, g["large_esm"].to(DEV), g["large_mol"].to(DEV))
        assert logits.dtype == dtype
        rows = g["large_rows"].to(DEV)
        assert rel_err(logits[:, rows].float(), g["large_logits_rows"]) < tol
        assert rel_err(logits.double().sum(-1).float(), g["large_logits_rowsum"]) < tol
        se = graph.score_entropy(logits, sigma[:, None], xt, x0)
        assert rel_err(se, g["large_se"]) < tol
        assert_close((dsigma[:, None] * se).mean(), g["large_loss"], tol, 0, "loss")
