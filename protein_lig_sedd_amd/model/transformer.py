"""Drop-in for protlig_dd/model/transformer.py: SEDD / _SEDD, DDiTBlock, DDitFinalLayer, TimestepEmbedder,
EmbeddingLayer, LayerNorm with the reference's constructor signatures, forward signatures, parameter names
(state-dict compatible) and initialisation — every tensor op runs in the HIP kernels of libsedd_hip.

The flash-attn call of the reference (transformer.py:180) is replaced by the library's own MFMA attention; there is
no flash_attn / Triton / TorchScript dependency.  `compute_dtype` (default torch.bfloat16) plays the role of the
reference's `autocast(dtype=torch.bfloat16)` region (transformer.py:306,373): Linear operands and activations use it,
LayerNorm statistics, the residual stream, rotary tables and all reductions stay fp32.  torch.float32 selects the
exact-fp32 MFMA path (what the reference computes on a machine without CUDA autocast; used for 1e-4 parity).
"""
import math

import torch
import torch.nn as nn

from .. import ops
from . import layers as Lyr
from .rotary import Rotary


_SPLIT = {}


def _params(module, names):
    """{reference parameter name: Parameter} of a module, through the modules' own dictionaries (nn.Module.__getattr__ costs
    ~1.5 us a hop: 45 us per block and call, which a launch-bound step pays 27 times)."""
    out = {}
    for n in names:
        parts = _SPLIT.get(n)
        if parts is None:
            parts = _SPLIT[n] = n.split(".")
        obj = module
        try:
            for part in parts[:-1]:
                obj = obj._modules[part]
            out[n] = obj._parameters[parts[-1]]
        except KeyError:  # not a registered sub-module / parameter: the general route
            obj = module
            for part in parts:
                obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
            out[n] = obj
    return out


def set_training(model, mode=True):
    """model.train(mode) only if some module is not in that mode already: the recursive setter costs ~1.8 ms on the
    dual-track model (478 modules, nn.Module.__setattr__ each), the check 0.1 ms — and the trainers call it every step, as
    the reference does."""
    if model.training != mode or any(m.training != mode for m in model.modules()):
        model.train(mode)
    return model


def _ret(grads, names):
    return tuple(grads.get(n) for n in names)


class LayerNorm(nn.Module):
    """transformer.py:30-38 (fp32 statistics, scale only)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones([dim]))
        self.dim = dim

    def forward(self, x):
        shp = x.shape
        y, _, _ = ops.ln_modulate_fwd(x.float().reshape(-1, 1, self.dim).contiguous(), self.weight.detach())
        return y.view(shp)


class _SigmaMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigma, mod, outer_silu, save, *params):
        P = dict(zip(Lyr.SIGMA_PARAMS, params))
        out, st = Lyr.sigma_map_forward(sigma, P, mod.frequency_embedding_size, outer_silu, save)
        ctx.st, ctx.P = st, P
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = Lyr.sigma_map_backward(dout, ctx.st, ctx.P)
        ctx.st = None
        return (None, None, None, None) + _ret(grads, Lyr.SIGMA_PARAMS)


class TimestepEmbedder(nn.Module):
    """transformer.py:56-94. `forward(t)` matches the reference; `forward(t, silu=True)` fuses the caller's F.silu."""

    def __init__(self, hidden_size, frequency_embedding_size=256, silu=True):
        super().__init__()
        self.mlp = nn.Sequential(
            nn.Linear(frequency_embedding_size, hidden_size, bias=True),
            nn.SiLU(),
            nn.Linear(hidden_size, hidden_size, bias=True),
        )
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        if max_period != 10000 or dim % 2:
            raise NotImplementedError("timestep_embedding kernel: max_period 10000 and even dim only")
        return ops.timestep_embedding(t, dim)

    def forward(self, t, silu=False):
        P = _params(self, Lyr.SIGMA_PARAMS)
        return _SigmaMapFn.apply(t, self, silu, torch.is_grad_enabled(), *[P[n] for n in Lyr.SIGMA_PARAMS])


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table):
        ctx.save_for_backward(idx)
        ctx.table = table
        return ops.embedding_fwd(table.detach(), idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        table = ctx.table
        if table.grad is not None:
            ops.embedding_bwd(dout, idx, table.grad)
            return None, None
        dt = torch.zeros_like(table)
        ops.embedding_bwd(dout, idx, dt)
        return None, dt


class EmbeddingLayer(nn.Module):
    """transformer.py:193-218"""

    def __init__(self, dim, vocab_dim):
        super().__init__()
        self.embedding = nn.Embedding(vocab_dim, dim)

    def forward(self, x):
        return _EmbeddingFn.apply(x, self.embedding.weight)


class _DDiTBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, blk, cos, sin, save, cu, links, *params):
        P = dict(zip(Lyr.BLOCK_PARAMS, params))
        p = blk.dropout if blk.training else 0.0
        out, st = Lyr.ddit_block_forward(x.contiguous(), c.contiguous(), P, blk.n_heads, cos, sin, p, blk.compute_dtype, save,
                                         cu_seqlens=cu, link_in=links[0], link_out=links[1])
        ctx.st, ctx.P, ctx.n_heads, ctx.blk = st, P, blk.n_heads, blk
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, dc, grads = Lyr.ddit_block_backward(dout.contiguous(), ctx.st, ctx.P, ctx.n_heads)
        ctx.st = None
        cb = getattr(ctx.blk, "_on_grads_ready", None)  # trainer hook: this block's gradient bucket is complete
        if cb is not None:
            cb()
        return (dx, dc, None, None, None, None, None, None) + _ret(grads, Lyr.BLOCK_PARAMS)


class DDiTBlock(nn.Module):
    """transformer.py:118-189.  Same parameters / init; forward(x, rotary_cos_sin, c, seqlens=None)."""

    def __init__(self, dim, n_heads, cond_dim, mlp_ratio=4, dropout=0.1, compute_dtype=torch.bfloat16):
        super().__init__()
        self.n_heads = n_heads
        self.norm1 = LayerNorm(dim)
        self.attn_qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.attn_out = nn.Linear(dim, dim, bias=False)
        self.dropout1 = nn.Dropout(dropout)
        self.norm2 = LayerNorm(dim)
        self.mlp = nn.Sequential(
            nn.Linear(dim, mlp_ratio * dim, bias=True),
            nn.GELU(approximate="tanh"),
            nn.Linear(mlp_ratio * dim, dim, bias=True),
        )
        self.dropout2 = nn.Dropout(dropout)
        self.dropout = dropout
        self.adaLN_modulation = nn.Linear(cond_dim, 6 * dim, bias=True)
        self.adaLN_modulation.weight.data.zero_()
        self.adaLN_modulation.bias.data.zero_()
        self.compute_dtype = compute_dtype
        self._rot = (None, None, None)

    def _tables(self, rotary_cos_sin):
        cos, sin = rotary_cos_sin
        if getattr(cos, "_sedd_compact", None) is not None:
            return cos._sedd_compact
        key = (cos.data_ptr(), cos.shape[1])
        if self._rot[0] != key:
            self._rot = (key,) + Lyr.compact_rotary(cos, sin)
        return self._rot[1], self._rot[2]

    def forward(self, x, rotary_cos_sin, c, seqlens=None, _links=(None, None)):
        """seqlens (transformer.py:173-179): the flash-attn varlen hook — `cu_seqlens = seqlens.cumsum(-1)` cuts the
        flattened (batch x sequence) token axis into segments and attention runs inside each segment (variable-length
        packing); as in the reference the caller's `seqlens` therefore starts with 0 and sums to batch x seq_len, and the
        padded row length is the stated maximum segment length.  None: one segment per row."""
        cos, sin = self._tables(rotary_cos_sin)
        P = _params(self, Lyr.BLOCK_PARAMS)
        cu = None if seqlens is None else seqlens.cumsum(-1).to(torch.int32).contiguous()
        return _DDiTBlockFn.apply(x, c, self, cos, sin, torch.is_grad_enabled(), cu, _links if cu is None else (None, None),
                                  *[P[n] for n in Lyr.BLOCK_PARAMS])


class _FinalLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, lyr, save, link, *params):
        P = dict(zip(Lyr.FINAL_PARAMS, params))
        out, st = Lyr.final_layer_forward(x.contiguous(), c.contiguous(), P, lyr.compute_dtype, save, link_in=link)
        ctx.st, ctx.P = st, P
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, dc, grads = Lyr.final_layer_backward(dout.contiguous(), ctx.st, ctx.P)
        ctx.st = None
        return (dx, dc, None, None, None) + _ret(grads, Lyr.FINAL_PARAMS)


class DDitFinalLayer(nn.Module):
    """transformer.py:221-238"""

    def __init__(self, hidden_size, out_channels, cond_dim, compute_dtype=torch.bfloat16):
        super().__init__()
        self.norm_final = LayerNorm(hidden_size)
        self.linear = nn.Linear(hidden_size, out_channels)
        self.linear.weight.data.zero_()
        self.linear.bias.data.zero_()
        self.adaLN_modulation = nn.Linear(cond_dim, 2 * hidden_size, bias=True)
        self.adaLN_modulation.weight.data.zero_()
        self.adaLN_modulation.bias.data.zero_()
        self.compute_dtype = compute_dtype

    def forward(self, x, c, _link=None):
        P = _params(self, Lyr.FINAL_PARAMS)
        return _FinalLayerFn.apply(x, c, self, torch.is_grad_enabled(), _link, *[P[n] for n in Lyr.FINAL_PARAMS])


class _EpilogueFn(torch.autograd.Function):
    """transformer.py:313-322: subtract log(E(sigma)) + log(V-1), zero the input token's column."""

    @staticmethod
    def forward(ctx, logits, sigma, indices, scale_by_sigma):
        ctx.save_for_backward(indices)
        return ops.logits_epilogue_(logits.contiguous().clone(), sigma, indices, scale_by_sigma)

    @staticmethod
    def backward(ctx, dout):
        (indices,) = ctx.saved_tensors
        return ops.logits_epilogue_bwd_(dout.contiguous().clone(), indices), None, None, None


def _cfg(config, path, default=None):
    obj = config
    for part in path.split("."):
        if isinstance(obj, dict):
            if part not in obj:
                return default
            obj = obj[part]
        else:
            if not hasattr(obj, part):
                return default
            obj = getattr(obj, part)
    return obj


class _SEDD(nn.Module):
    """transformer.py:332-392 (the plain upstream SEDD). forward(indices, sigma) -> log-score [B, L, V]."""

    def __init__(self, config, compute_dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.compute_dtype = compute_dtype
        self.absorb = _cfg(config, "graph.type") == "absorb"
        vocab_size = _cfg(config, "tokens") + (1 if self.absorb else 0)
        hidden, cond = _cfg(config, "model.hidden_size"), _cfg(config, "model.cond_dim")
        heads = _cfg(config, "model.n_heads")
        self.vocab_embed = EmbeddingLayer(hidden, vocab_size)
        self.sigma_map = TimestepEmbedder(cond)
        self.rotary_emb = Rotary(hidden // heads)
        self.blocks = nn.ModuleList([
            DDiTBlock(hidden, heads, cond, dropout=_cfg(config, "model.dropout"), compute_dtype=compute_dtype)
            for _ in range(_cfg(config, "model.n_blocks"))
        ])
        self.output_layer = DDitFinalLayer(hidden, vocab_size, cond, compute_dtype=compute_dtype)
        self.scale_by_sigma = _cfg(config, "model.scale_by_sigma")

    def set_compute_dtype(self, dtype):
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, (DDiTBlock, DDitFinalLayer)):
                m.compute_dtype = dtype
        return self

    def _cond(self, sigma, esm_cond=None, mol_cond=None):
        return self.sigma_map(sigma, silu=True)

    def forward(self, indices, sigma):
        return self._forward(indices, sigma)

    def _forward(self, indices, sigma, esm_cond=None, mol_cond=None):
        x = self.vocab_embed(indices)
        c = self._cond(sigma, esm_cond, mol_cond)
        rotary_cos_sin = self.rotary_emb(x)
        # consecutive blocks share a BlockLink: the residual-stream passes at their boundary run fused (layers.BlockLink)
        n = len(self.blocks)
        nxt = list(self.blocks[1:]) + [self.output_layer]
        links = [Lyr.BlockLink(self.blocks[i], nxt[i]) for i in range(n)] if Lyr.BLOCK_CHAIN else [None] * n
        for i, blk in enumerate(self.blocks):
            x = blk(x, rotary_cos_sin, c, seqlens=None, _links=(links[i - 1] if i > 0 else None, links[i]))
        x = self.output_layer(x, c, _link=links[n - 1] if n else None)
        if self.scale_by_sigma:
            assert self.absorb, "Haven't configured this to work."
        return _EpilogueFn.apply(x, sigma, indices, bool(self.scale_by_sigma))


COND_PARAMS = ("proj.weight", "proj.bias", "norm.weight", "norm.bias")


class _PooledCondFn(torch.autograd.Function):
    """transformer.py:289-301: c' = mean([c, esm_norm(esm_proj(esm_cond)), mol_norm(mol_proj(mol_cond))]) over the terms
    that are given.  The projections are fp32 Linears (they sit outside the reference's autocast region) on the MFMA
    GEMM; both BatchNorm1d's (batch or running statistics, running-stat update) and the mean are one kernel."""

    @staticmethod
    def forward(ctx, c, esm_cond, mol_cond, model, save, *params):
        terms, lin_in, mods = [], [], []
        for k, (cond, proj, norm) in enumerate(((esm_cond, model.esm_proj, getattr(model, "esm_norm", None)),
                                                (mol_cond, model.mol_proj, getattr(model, "mol_norm", None)))):
            if cond is None or proj is None:
                terms.append(None)
                lin_in.append(None)
                mods.append(None)
                continue
            xin = cond.detach().float().contiguous()
            lin = Lyr.linear_fwd(xin, proj.weight, proj.bias)
            terms.append((lin, norm.weight.detach(), norm.bias.detach(), norm.running_mean, norm.running_var))
            lin_in.append(xin)
            mods.append((proj, norm))
        training = next((m[1].training for m in mods if m is not None), model.training)
        momentum = 0.1
        for m in mods:
            if m is not None:
                momentum = m[1].momentum if m[1].momentum is not None else 0.1
        out, saved = ops.cond_combine_fwd(c.contiguous(), terms, training, momentum,
                                          next((m[1].eps for m in mods if m is not None), 1e-5))
        if training:
            for m in mods:
                if m is not None and m[1].num_batches_tracked is not None:
                    m[1].num_batches_tracked.add_(1)
        ctx.saved, ctx.lin_in, ctx.mods, ctx.training = (saved, lin_in, mods, training) if save else (None,) * 4
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, lin_in, mods, training = ctx.saved, ctx.lin_in, ctx.mods, ctx.training
        weights = [None if m is None else m[1].weight.detach() for m in mods]
        sinks = [None if m is None or m[1].weight.grad is None or m[1].bias.grad is None
                 else (m[1].weight.grad, m[1].bias.grad) for m in mods]
        if any(s is None for s, m in zip(sinks, mods) if m is not None):
            sinks = [None, None]
        dc, dxs, dws = ops.cond_combine_bwd(dout.contiguous(), saved, weights, sinks, training)
        grads = []
        for k, m in enumerate(mods):
            if m is None:
                grads += [None] * 4
                continue
            proj, norm = m
            _, dW, db = Lyr.linear_bwd(lin_in[k], dxs[k], proj.weight, proj.bias, need_dx=False)
            dnw, dnb = dws[k] if dws[k] is not None else (None, None)
            grads += [dW, db, dnw, dnb]
        ctx.saved = ctx.lin_in = None
        return (dc, None, None, None, None) + tuple(grads)


class SEDD(_SEDD):
    """transformer.py:240-325: _SEDD plus the pooled ESM / MolFormer conditioners (esm_proj, mol_proj and BatchNorm1d,
    averaged with the sigma embedding, :289-301) — what run_train_ddp_protlig_transf.py:284 calls with esm_cond /
    mol_cond.  State-dict keys as the reference: esm_proj.*, mol_proj.*, esm_norm.*, mol_norm.* (incl. running stats)."""

    def __init__(self, config, compute_dtype=torch.bfloat16):
        super().__init__(config, compute_dtype)
        cond = _cfg(config, "model.cond_dim")
        esm_dim, mol_dim = _cfg(config, "model.esm_dim"), _cfg(config, "model.molformer_dim")
        if esm_dim:
            self.esm_proj = nn.Linear(esm_dim, cond)
            self.esm_norm = nn.BatchNorm1d(cond)
        else:
            self.esm_proj = None
        if mol_dim:
            self.mol_proj = nn.Linear(mol_dim, cond)
            self.mol_norm = nn.BatchNorm1d(cond)
        else:
            self.mol_proj = None

    def forward(self, indices, sigma, esm_cond=None, mol_cond=None):
        return self._forward(indices, sigma, esm_cond, mol_cond)

    def _cond(self, sigma, esm_cond=None, mol_cond=None):
        c = self.sigma_map(sigma, silu=True)
        use_e = self.esm_proj is not None and esm_cond is not None
        use_m = self.mol_proj is not None and mol_cond is not None
        if not (use_e or use_m):
            return c
        params = []
        for use, proj, norm in ((use_e, self.esm_proj, getattr(self, "esm_norm", None)),
                                (use_m, self.mol_proj, getattr(self, "mol_norm", None))):
            params += [proj.weight, proj.bias, norm.weight, norm.bias] if use else [None] * 4
        return _PooledCondFn.apply(c, esm_cond if use_e else None, mol_cond if use_m else None, self,
                                   torch.is_grad_enabled(), *params)
