"""CPU oracle: a plain-PyTorch (CPU, fp32) restatement of the SEDD hot path of architvasan/protein_lig_sedd.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / the timed CPU baseline; the product path (protein_lig_sedd_amd) never does and
fails loudly when the HIP library is missing.

Pinned: every function here is checked against golden vectors produced by importing the reference itself in the
build container (oracle/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_vs_golden.py).  The one boundary
the reference does not pin is flash-attn (external wheel, version unpinned, no reference test): it is defined
here as exact fp32-accumulated softmax attention ("parity unpinned" at that boundary).

Everything is functional over a state dict that uses the reference's parameter names, so the same weights load
into the reference modules, this oracle and the HIP-backed modules.  Citations are to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------- noise
# protlig_dd/processing/noise_lib.py:45-94


def loglinear_noise(t, eps=1e-3):
    """noise_lib.py:72-76 -> (sigma, dsigma)"""
    return -torch.log1p(-(1 - eps) * t), (1 - eps) / (1 - (1 - eps) * t)


def geometric_noise(t, sigma_min=1e-3, sigma_max=1.0):
    """noise_lib.py:53-57"""
    s0, s1 = torch.tensor(float(sigma_min)), torch.tensor(float(sigma_max))
    sig = s0 ** (1 - t) * s1 ** t
    return sig, sig * (s1.log() - s0.log())


def cosine_noise(t, sigma_min=1e-3, sigma_max=1.0, T=1.0):
    """noise_lib.py:86-94"""
    s0, s1 = torch.tensor(float(sigma_min)), torch.tensor(float(sigma_max))
    sig = s1 + 0.5 * (s0 - s1) * (1 + torch.cos(math.pi * t / T))
    dsig = (math.pi / (2 * T)) * (s1 - s0) * torch.sin(math.pi * t / T)
    return sig, dsig


def make_noise(kind, **kw):
    fn = {"loglinear": loglinear_noise, "geometric": geometric_noise, "cosine": cosine_noise}[kind]
    return lambda t: fn(t, **kw)


# --------------------------------------------------------------------------------------------- graph
def esigm1(sigma):
    """graph_lib.py:279-283 / transformer.py:315"""
    return torch.where(sigma < 0.5, torch.expm1(sigma), torch.exp(sigma) - 1)


def gumbel_argmax(p, u):
    """utils/catsample.py:12-13 with the uniform noise passed in. First-index tie-break (torch CPU argmax)."""
    return (p / (1e-10 - (u + 1e-10).log())).argmax(dim=-1)


class AbsorbingCPU:
    """graph_lib.py:200-314.  dim = tokens + 1, mask id = dim - 1."""

    absorb = True

    def __init__(self, tokens):
        self.dim = tokens + 1

    def sample_transition(self, i, sigma, u):                                   # :258-262
        return torch.where(u < 1 - (-sigma).exp(), torch.full_like(i, self.dim - 1), i)

    @staticmethod
    def curriculum_factor(step, preschool=100, elementary=500, middle=1000, high=1500, adult=2000):  # :236-249
        if step < preschool:
            return 3
        if step < elementary:
            return 3 - min(1.0, step / elementary)
        if step < middle:
            return 2
        if step < high:
            return 2 - min(1.0, step / high) * 0.8
        if step < adult:
            return 1.2
        return 1

    def sample_transition_curriculum(self, i, sigma, step, u):                  # :236-256
        return self.sample_transition(i, sigma * self.curriculum_factor(step), u)

    def rate(self, i):                                                          # :213-216
        V = self.dim
        return F.one_hot(torch.full_like(i, V - 1), V) - F.one_hot(i, V)

    def transp_rate(self, i):                                                   # :218-221
        e = -F.one_hot(i, self.dim)
        return e + (i == self.dim - 1)[..., None].long()

    def reverse_rate(self, i, score):                                           # :79-93
        r = self.transp_rate(i) * score
        r = r.scatter(-1, i[..., None], torch.zeros_like(r[..., :1]))
        return r.scatter(-1, i[..., None], -r.sum(-1, keepdim=True))

    def sample_rate(self, i, rate, u):                                          # :95-96
        return gumbel_argmax(F.one_hot(i, self.dim).to(rate) + rate, u)

    def staggered_score(self, score, dsigma):                                   # :264-269 (dsigma is [B,1])
        extra = (1 - dsigma.exp()) * score.sum(-1)
        out = score * dsigma.exp()[:, None]
        out[..., -1] = out[..., -1] + extra
        return out

    def transp_transition(self, i, sigma):                                      # :226-234
        sig = sigma.view(*sigma.shape, *((1,) * (i.dim() + 1 - sigma.dim())))
        e = (-sig).exp() * F.one_hot(i, self.dim)
        return e + torch.where(i == self.dim - 1, 1 - (-sig).squeeze(-1).exp(), 0.0)[..., None]

    def sample_limit(self, *dims):                                              # :271-272
        return torch.full(dims, self.dim - 1, dtype=torch.int64)

    def score_entropy(self, score, sigma, x, x0):                               # :274-314
        V = self.dim
        masked = x == V - 1
        r = 1.0 / esigm1(sigma).expand_as(x)
        s = score.float()
        pos = s[..., :-1].exp().sum(-1)
        neg = r * s.gather(-1, x0.clamp(0, V - 1)[..., None]).squeeze(-1)
        const = r * (r.log() - 1)
        return torch.where(masked, pos - neg + const, torch.zeros_like(pos))


class UniformCPU:
    """graph_lib.py:124-197."""

    absorb = False

    def __init__(self, dim):
        self.dim = dim

    def rate(self, i):                                                          # :139-142
        e = torch.ones(*i.shape, self.dim) / self.dim
        return e.scatter(-1, i[..., None], -(self.dim - 1) / self.dim)

    transp_rate = rate

    def transition(self, i, sigma):                                             # :147-151
        tr = torch.ones(*i.shape, self.dim) * (1 - (-sigma[..., None]).exp()) / self.dim
        tr = tr.scatter(-1, i[..., None], torch.zeros_like(tr))
        return tr.scatter(-1, i[..., None], 1 - tr.sum(-1, keepdim=True))

    transp_transition = transition

    def sample_transition(self, i, sigma, u, r):                                # :156-160
        return torch.where(u < 1 - (-sigma).exp(), r, i)

    def reverse_rate(self, i, score):
        rr = self.transp_rate(i) * score
        rr = rr.scatter(-1, i[..., None], torch.zeros_like(rr[..., :1]))
        return rr.scatter(-1, i[..., None], -rr.sum(-1, keepdim=True))

    def sample_rate(self, i, rate, u):
        return gumbel_argmax(F.one_hot(i, self.dim).to(rate) + rate, u)

    def staggered_score(self, score, dsigma):                                   # :162-165
        V = score.shape[-1]
        w = (-dsigma).exp()[..., None]
        return ((w - 1) / (V * w)) * score.sum(-1, keepdim=True) + score / w

    def score_entropy(self, score, sigma, x, x0):                               # :170-197
        V = self.dim
        e = esigm1(sigma)
        ratio = 1 - V / (e + V)
        s = score.float()
        sx = s.gather(-1, x[..., None]).squeeze(-1)
        neg = s.mean(-1) - sx / V
        neg = torch.where(x == x0, ratio * neg, s.gather(-1, x0[..., None]).squeeze(-1) / e + neg)
        const = torch.where(x == x0, (V - 1) / V * ratio * (ratio.log() - 1),
                            ((-ratio.log() - 1) / ratio - (V - 2)) / V)
        sexp = s.exp()
        pos = sexp.mean(-1) - sexp.gather(-1, x[..., None]).squeeze(-1) / V
        return pos - neg + const


def make_graph(kind, tokens):
    return AbsorbingCPU(tokens) if kind == "absorb" else UniformCPU(tokens)


# --------------------------------------------------------------------------------------------- predictors
def euler_step(graph, logits, x, sigma_dsigma, dt, u):
    """sampling.py:61-80 given the model's log-score. sigma_dsigma = noise(t) with t [B,1]."""
    _, dsigma = sigma_dsigma
    score = logits.float().exp()
    rev = dt * dsigma[..., None] * graph.reverse_rate(x, score)
    return graph.sample_rate(x, rev, u)


def analytic_step(graph, logits, x, dsigma, u):
    """sampling.py:90-99; dsigma = sigma(t) - sigma(t - dt), shape [B,1]."""
    score = logits.float().exp()
    probs = graph.staggered_score(score, dsigma) * graph.transp_transition(x, dsigma)
    return gumbel_argmax(probs, u)


def denoise_step(graph, logits, x, sigma, u):
    """sampling.py:107-132; sigma = sigma(t_eps) [B,1]; u has V-1 columns when the graph is absorbing."""
    score = logits.float().exp()
    probs = graph.staggered_score(score, sigma) * graph.transp_transition(x, sigma)
    if graph.absorb:
        probs = probs[..., :-1]
    return gumbel_argmax(probs, u)


# --------------------------------------------------------------------------------------------- model pieces
def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11, section 3.3 and
    the Random123 reference constants) on numpy uint32 arrays: counter [..., 4], key [..., 2] -> [..., 4].  No
    counterpart in the reference (it draws torch.rand_like, utils/catsample.py:12); this restates the published
    generator that sedd_predictor_step_philox runs in-kernel."""
    import numpy as np
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), \
        np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, -1).astype(np.uint32)


def philox_uniform(seed, offset, M, Vu):
    """The [M, Vu] uniforms of sedd_predictor_step_philox (include/sedd_hip.h): word (j & 3) of
    philox(counter = {row_lo, row_hi, j >> 2, offset}, key = {seed_lo, seed_hi}); u = (word >> 8) * 2^-24."""
    import numpy as np
    nq = (Vu + 3) // 4
    rows = np.arange(M, dtype=np.uint64)[:, None].repeat(nq, 1)
    ctr = np.stack([rows & np.uint64(0xFFFFFFFF), rows >> np.uint64(32),
                    np.arange(nq, dtype=np.uint64)[None, :].repeat(M, 0), np.full((M, nq), offset, dtype=np.uint64)], -1)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint64), (M, nq, 2))
    w = philox4x32_10(ctr, key).reshape(M, nq * 4)[:, :Vu]
    return torch.from_numpy(((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)))


def timestep_embedding(sigma, dim=256, max_period=10000):
    """transformer.py:70-89"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = sigma[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def sigma_cond(sd, sigma, prefix="sigma_map."):
    """F.silu(TimestepEmbedder(sigma)) - transformer.py:91-94,285"""
    h = F.linear(timestep_embedding(sigma), sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"])
    h = F.linear(F.silu(h), sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"])
    return F.silu(h)


def rotary_cos_sin(L, d, base=10000.0):
    """rotary.py:13-27: cos/sin [L, d] (NeoX halves duplicated)."""
    inv = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    fr = torch.outer(torch.arange(L).float(), inv)
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def apply_rotary(qk, cos, sin):
    """rotary.py:30-39 on [B, L, H, d] with cos/sin [L, d]."""
    d = qk.shape[-1]
    rot = torch.cat([-qk[..., d // 2:], qk[..., :d // 2]], -1)
    return qk * cos[None, :, None, :] + rot * sin[None, :, None, :]


def layer_norm_w(x, w):
    """transformer.py:35-38"""
    return F.layer_norm(x.float(), [x.shape[-1]]) * w[None, None, :]


def _cast(x, bf16):
    return x.to(torch.bfloat16) if bf16 else x


def _linear(x, w, b, bf16):
    """nn.Linear under autocast(bf16) rounds inputs, weights and the output to bf16."""
    if bf16:
        return F.linear(x.to(torch.bfloat16), w.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16))
    return F.linear(x, w, b)


def softmax_attention(q, k, v, key_mask=None):
    """Exact attention on [B, L, H, d] tensors (the flash-attn boundary, transformer.py:180-181): fp32 softmax of
    q k^T d^-1/2, no dropout, optional key mask [B, Lk] (True = keep)."""
    d = q.shape[-1]
    s = torch.einsum("blhd,bmhd->bhlm", q.float(), k.float()) * (d ** -0.5)
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    return torch.einsum("bhlm,bmhd->blhd", p, v.float()).to(q.dtype)


def varlen_attention(q, k, v, cu):
    """flash_attn_varlen_qkvpacked_func in its general form (transformer.py:173-181): q / k / v [T, H, d] (the packed
    token axis), cu = cumulative segment offsets starting at 0; exact softmax attention inside each segment; tokens
    outside every segment are zero."""
    out = torch.zeros_like(q, dtype=torch.float32)
    d = q.shape[-1]
    for a, b in zip(cu[:-1], cu[1:]):
        if b > a:
            sc = torch.einsum("lhd,mhd->hlm", q[a:b].float(), k[a:b].float()) * (d ** -0.5)
            out[a:b] = torch.einsum("hlm,mhd->lhd", torch.softmax(sc, -1), v[a:b].float())
    return out.to(q.dtype)


def ddit_block(sd, pre, x, c, n_heads, bf16=False, seqlens=None):
    """DDiTBlock.forward, flash variant (transformer.py:153-189), dropout = 0 / eval.  seqlens: the varlen hook
    (:173-179), cu_seqlens = seqlens.cumsum(-1) over the flattened (batch x sequence) token axis."""
    B, L, D = x.shape
    d = D // n_heads
    mod = _linear(c, sd[pre + "adaLN_modulation.weight"], sd[pre + "adaLN_modulation.bias"], bf16)
    # modulate_fused / bias_dropout_add_scale_fused are TorchScript-fused in the reference (fused_add_dropout_scale.py:
    # 7-10 turns the fusers on): inside a fused kernel bf16 operands are widened once and the arithmetic stays fp32,
    # so `1 + scale` and `scale * x` are NOT rounded to bf16 on the way (pinned by golden g17: the reference under bf16
    # autocast)
    sh1, sc1, g1, sh2, sc2, g2 = mod.float()[:, None].chunk(6, dim=2)
    h = layer_norm_w(x, sd[pre + "norm1.weight"]) * (1 + sc1) + sh1
    qkv = _linear(h, sd[pre + "attn_qkv.weight"], None, bf16).view(B, L, 3, n_heads, d)
    cos, sin = rotary_cos_sin(L, d)
    cos, sin = _cast(cos, bf16), _cast(sin, bf16)
    q, k, v = qkv.unbind(2)
    q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
    if seqlens is None:
        o = softmax_attention(q, k, v).reshape(B, L, D)
    else:
        cu = [int(t) for t in torch.as_tensor(seqlens).cumsum(-1).tolist()]
        o = varlen_attention(q.reshape(B * L, n_heads, d), k.reshape(B * L, n_heads, d), v.reshape(B * L, n_heads, d),
                             cu).reshape(B, L, D)
    x = x + g1 * _linear(o, sd[pre + "attn_out.weight"], None, bf16).float()
    h = layer_norm_w(x, sd[pre + "norm2.weight"]) * (1 + sc2) + sh2
    h = _linear(h, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"], bf16)
    h = _linear(F.gelu(h, approximate="tanh"), sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"], bf16)
    return x + g2 * h.float()


def final_layer(sd, pre, x, c, bf16=False):
    """DDitFinalLayer.forward (transformer.py:234-238)"""
    mod = _linear(c, sd[pre + "adaLN_modulation.weight"], sd[pre + "adaLN_modulation.bias"], bf16)
    sh, sc = mod.float()[:, None].chunk(2, dim=2)
    h = layer_norm_w(x, sd[pre + "norm_final.weight"]) * (1 + sc) + sh
    return _linear(h, sd[pre + "linear.weight"], sd[pre + "linear.bias"], bf16)


def logits_epilogue(x, sigma, indices, scale_by_sigma=True):
    """transformer.py:313-322"""
    if scale_by_sigma:
        x = x - esigm1(sigma).log().to(x.dtype)[:, None, None] - math.log(x.shape[-1] - 1)
    return torch.scatter(x, -1, indices[..., None], torch.zeros_like(x[..., :1]))


def batchnorm1d(x, w, b, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d on [B, C] (transformer.py:259-260,292,298).  training: batch statistics (biased variance in the
    normalisation, unbiased in the running update); eval: the running statistics.  Returns (y, new running_mean, new
    running_var); the inputs are not modified."""
    if training:
        if x.shape[0] < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        n = x.shape[0]
        rm = (1 - momentum) * running_mean + momentum * mean.detach()
        rv = (1 - momentum) * running_var + momentum * var.detach() * (n / (n - 1))
    else:
        mean, var, rm, rv = running_mean, running_var, running_mean, running_var
    return (x - mean) / torch.sqrt(var + eps) * w + b, rm, rv


def sedd_conditioner(sd, c, esm_cond=None, mol_cond=None, training=False, bn_state=None):
    """SEDD.forward's pooled conditioners (transformer.py:289-301): c' = mean([c, BN(esm_proj(esm)), BN(mol_proj(mol))])
    over the terms that are given; fp32 (outside the autocast region).  bn_state: dict of running statistics, updated
    in place in training mode (default: the entries of sd)."""
    st = sd if bn_state is None else bn_state
    conds = [c]
    for cond, proj, norm in ((esm_cond, "esm_proj", "esm_norm"), (mol_cond, "mol_proj", "mol_norm")):
        if cond is None:
            continue
        h = F.linear(cond, sd[proj + ".weight"], sd[proj + ".bias"])
        y, rm, rv = batchnorm1d(h, sd[norm + ".weight"], sd[norm + ".bias"], st[norm + ".running_mean"],
                                st[norm + ".running_var"], training)
        if training:
            st[norm + ".running_mean"], st[norm + ".running_var"] = rm, rv
        conds.append(y)
    return torch.mean(torch.stack(conds), dim=0)


def sedd_forward(sd, indices, sigma, n_heads, n_blocks, scale_by_sigma=True, bf16=False, esm_cond=None, mol_cond=None,
                 training=False, bn_state=None):
    """_SEDD.forward (transformer.py:365-392); with esm_cond / mol_cond: SEDD.forward (:277-325)."""
    x = F.embedding(indices, sd["vocab_embed.embedding.weight"])
    c = sigma_cond(sd, sigma)
    if esm_cond is not None or mol_cond is not None:
        c = sedd_conditioner(sd, c, esm_cond, mol_cond, training, bn_state)
    for i in range(n_blocks):
        x = ddit_block(sd, f"blocks.{i}.", x, c, n_heads, bf16)
    x = final_layer(sd, "output_layer.", x, c, bf16)
    return logits_epilogue(x, sigma, indices, scale_by_sigma)


def sedd_loss(sd, graph, noise, x0, t, xt, n_heads, n_blocks, bf16=False):
    """losses.py:11-30 with t and the perturbed batch given."""
    sigma, dsigma = noise(t)
    logits = sedd_forward(sd, xt, sigma, n_heads, n_blocks, True, bf16)
    se = graph.score_entropy(logits, sigma[:, None], xt, x0)
    return (dsigma[:, None] * se).mean(), logits


# --------------------------------------------------------------------------------------------- cross model
def _mha(sd, pre, q_in, k_in, v_in, n_heads, key_mask=None):
    """nn.MultiheadAttention(batch_first=True), eval: in_proj on q,k,v separately, softmax attention, out_proj."""
    D = q_in.shape[-1]
    w, b = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    q = F.linear(q_in, w[:D], b[:D])
    k = F.linear(k_in, w[D:2 * D], b[D:2 * D])
    v = F.linear(v_in, w[2 * D:], b[2 * D:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    d = D // n_heads
    o = softmax_attention(q.view(B, Lq, n_heads, d), k.view(B, Lk, n_heads, d), v.view(B, Lk, n_heads, d), key_mask)
    return F.linear(o.reshape(B, Lq, D), sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def cross_attention_layer(sd, pre, x, ctx, n_heads, ctx_mask=None):
    """CrossAttentionLayer.forward (transformers_protlig_cross.py:117-140), dropout = 0."""
    xn = F.layer_norm(x, [x.shape[-1]], sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return x + _mha(sd, pre + "cross_attn.", xn, ctx, ctx, n_heads, ctx_mask)


def cross_ddit_block(sd, pre, x, c, n_heads, other=None, other_mask=None):
    """DDiTBlock.forward, cross variant (transformers_protlig_cross.py:279-320), as written: the 'self-attention'
    MHA receives [L, B, D] tensors in a batch_first module, so it attends over the batch axis."""
    B, L, D = x.shape
    d = D // n_heads
    mod = F.linear(c, sd[pre + "adaLN_modulation.weight"], sd[pre + "adaLN_modulation.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = mod[:, None].chunk(6, dim=2)
    h = layer_norm_w(x, sd[pre + "norm1.weight"]) * (1 + sc1) + sh1
    qkv = F.linear(h, sd[pre + "attn_qkv.weight"], sd[pre + "attn_qkv.bias"]).view(B, L, 3, n_heads, d)
    cos, sin = rotary_cos_sin(L, d)
    q, k, v = qkv.unbind(2)
    q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
    # [B, L, H, d] -> [L, B, D]: MHA(batch_first) then sees batch = L, sequence = B
    qs, ks, vs = (z.reshape(B, L, D).transpose(0, 1) for z in (q, k, v))
    a = _mha(sd, pre + "attn.", qs, ks, vs, n_heads).transpose(0, 1)
    x = x + g1 * F.linear(a, sd[pre + "attn_out.weight"], sd[pre + "attn_out.bias"])
    if other is not None and (pre + "cross_attn.norm.weight") in sd:
        x = cross_attention_layer(sd, pre + "cross_attn.", x, other, n_heads, other_mask)
    h = layer_norm_w(x, sd[pre + "norm2.weight"]) * (1 + sc2) + sh2
    h = F.linear(h, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])
    h = F.linear(F.gelu(h, approximate="tanh"), sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"])
    return x + g2 * h


def cross_single_track(sd, indices, sigma, track, n_heads, n_blocks, other=None, other_mask=None):
    """DualTrackDiffusionTransformer.forward_single_track (:372-453); sd keys carry the 'transformer.' prefix."""
    P = "transformer."
    emb, blocks, outp = (("protein_embed", "blocks_prot", "protein_output") if track == "protein"
                         else ("ligand_embed", "blocks_lig", "ligand_output"))
    x = F.embedding(indices, sd[P + emb + ".embedding.weight"])
    c = sigma_cond(sd, sigma, P + "sigma_map.")
    for i in range(n_blocks):
        x = cross_ddit_block(sd, f"{P}{blocks}.{i}.", x, c, n_heads, other, other_mask)
    x = final_layer(sd, P + outp + ".", x, c)
    return logits_epilogue(x, sigma, indices, True)


def cross_context(sd, indices, sigma, track, n_heads, n_blocks):
    """Context for the *_given_* modes: first n_blocks//2 blocks of the conditioning track without cross-attention
    (transformers_protlig_cross.py:608-616, 636-642)."""
    P = "transformer."
    emb, blocks = ("protein_embed", "blocks_prot") if track == "protein" else ("ligand_embed", "blocks_lig")
    h = F.embedding(indices, sd[P + emb + ".embedding.weight"])
    c = sigma_cond(sd, sigma, P + "sigma_map.")
    for i in range(n_blocks // 2):
        h = cross_ddit_block(sd, f"{P}{blocks}.{i}.", h, c, n_heads)
    return h


def cross_joint(sd, prot_idx, lig_idx, sigma, n_heads, n_blocks, prot_mask=None, lig_mask=None):
    """forward_joint (:455-517): both tracks read the other's pre-update hidden state."""
    P = "transformer."
    hp = F.embedding(prot_idx, sd[P + "protein_embed.embedding.weight"])
    hl = F.embedding(lig_idx, sd[P + "ligand_embed.embedding.weight"])
    c = sigma_cond(sd, sigma, P + "sigma_map.")
    for i in range(n_blocks):
        np_ = cross_ddit_block(sd, f"{P}blocks_prot.{i}.", hp, c, n_heads, hl, lig_mask)
        nl_ = cross_ddit_block(sd, f"{P}blocks_lig.{i}.", hl, c, n_heads, hp, prot_mask)
        hp, hl = np_, nl_
    lp = final_layer(sd, P + "protein_output.", hp, c)
    ll = final_layer(sd, P + "ligand_output.", hl, c)
    return logits_epilogue(lp, sigma, prot_idx, True), logits_epilogue(ll, sigma, lig_idx, True)


# --------------------------------------------------------------------------------------------- optimiser / EMA
def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, clip_coef=1.0):
    """torch.optim.AdamW single-tensor update (the reference's optimizer, losses.py:39-41) with the gradient
    pre-multiplied by the clip coefficient (clip_grad_norm_, losses.py:65-66). In place; step is 1-based."""
    g = g * clip_coef
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_decay(num_updates, decay):
    """model/ema.py:40-41 (num_updates already incremented)."""
    return min(decay, (1 + num_updates) / (10 + num_updates))


def ema_update(shadow, p, decay_eff):
    """model/ema.py:44-52: s -= (1 - decay) * (s - p)"""
    shadow.sub_((1.0 - decay_eff) * (shadow - p))


def clip_coef(total_norm, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6))"""
    return min(1.0, max_norm / (float(total_norm) + 1e-6))
