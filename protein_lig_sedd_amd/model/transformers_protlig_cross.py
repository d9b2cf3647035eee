"""Drop-in for protlig_dd/model/transformers_protlig_cross.py — the dual-track (protein, ligand) DDiT with
cross-attention that `run_train_crossatt.sh` trains: `CrossAttentionLayer`, `DDiTBlock` (cross variant),
`DualTrackDiffusionTransformer`, `ProteinLigandDiffusionModel` with the reference's constructor / forward signatures,
modes (`protein_only`, `ligand_only`, `ligand_given_protein`, `protein_given_ligand`, `joint`), errors and state-dict
keys (`transformer.blocks_{prot,lig}.N.*`, `…attn.in_proj_{weight,bias}`, `…cross_attn.{cross_attn.*,norm.*}`).

The reference runs this model under `autocast(dtype=torch.float32)` (:401, :478), i.e. in fp32: `compute_dtype`
defaults to torch.float32 (exact-fp32 MFMA GEMM / attention); torch.bfloat16 is available as an opt-in.
The pretrained ESM / MolFormer encoders (:142-233) need network downloads and are out of scope.
"""
import torch
import torch.nn as nn

from .. import ops
from . import cross_layers as CL
from . import layers as Lyr
from .rotary import Rotary
from .transformer import (DDitFinalLayer, EmbeddingLayer as _Emb, LayerNorm, TimestepEmbedder, _EpilogueFn, _cfg,
                          _params, _ret)


class EmbeddingLayer(_Emb):
    """transformers_protlig_cross.py:79-85: nn.Embedding(vocab_dim + 1, dim)."""

    def __init__(self, dim, vocab_dim):
        super().__init__(dim, vocab_dim + 1)


class CrossAttentionLayer(nn.Module):
    """transformers_protlig_cross.py:107-140 (parameter container + stand-alone forward)."""

    def __init__(self, d_model, n_heads, dropout=0.1):
        super().__init__()
        self.cross_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.n_heads = n_heads

    def forward(self, x, context, x_mask=None, context_mask=None):
        """Inference-only stand-alone form (inside DDiTBlock the layer is fused into the block's fwd/bwd)."""
        B, L, D = x.shape
        H, d = self.n_heads, D // self.n_heads
        W, b = self.cross_attn.in_proj_weight, self.cross_attn.in_proj_bias
        xn, _, _ = ops.ln_modulate_fwd(x.float().contiguous(), self.norm.weight.detach(), bias=self.norm.bias.detach())
        cq = CL.part_fwd(xn.view(B * L, D), W, b, 0, D).view(B, L, H, d)
        Lo = context.shape[1]
        ckv = CL.part_fwd(context.float().contiguous().view(B * Lo, D), W, b, D, 3 * D).view(B, Lo, 2, H, d)
        co, _ = ops.attn_fwd(cq, ckv[:, :, 0], ckv[:, :, 1], key_mask=context_mask)
        xc = Lyr.linear_fwd(co.view(B * L, D), self.cross_attn.out_proj.weight, self.cross_attn.out_proj.bias)
        return x + xc.view(B, L, D)


class _CrossBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, other, other_mask, blk, cos, sin, save, names, *params):
        P = dict(zip(names, params))
        p = blk.dropout if blk.training else 0.0
        out, st = CL.cross_block_forward(x.contiguous(), c.contiguous(), other, other_mask, P, blk.n_heads, cos, sin, p,
                                         blk.compute_dtype, save, blk.seq_axis_attention)
        ctx.st, ctx.P, ctx.names, ctx.blk = st, P, names, blk
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, dc, d_other, grads = CL.cross_block_backward(dout.contiguous(), ctx.st, ctx.P, ctx.blk.n_heads)
        ctx.st = None
        cb = getattr(ctx.blk, "_on_grads_ready", None)
        if cb is not None:
            cb()
        return (dx, dc, d_other, None, None, None, None, None, None) + _ret(grads, ctx.names)


class DDiTBlock(nn.Module):
    """transformers_protlig_cross.py:237-320."""

    def __init__(self, dim, n_heads, dropout, cond_dim, mlp_ratio=4, enable_cross_attention=True,
                 compute_dtype=torch.float32, seq_axis_attention=False):
        super().__init__()
        self.dim, self.n_heads, self.dropout, self.cond_dim, self.mlp_ratio = dim, n_heads, dropout, cond_dim, mlp_ratio
        self.enable_cross_attention = enable_cross_attention
        self.norm1 = LayerNorm(dim)
        self.attn_qkv = nn.Linear(dim, 3 * dim)
        self.attn = nn.MultiheadAttention(dim, n_heads, dropout=dropout, batch_first=True)
        self.attn_out = nn.Linear(dim, dim)
        if enable_cross_attention:
            self.cross_attn = CrossAttentionLayer(dim, n_heads, dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm2 = LayerNorm(dim)
        self.mlp = nn.Sequential(
            nn.Linear(dim, dim * mlp_ratio, bias=True),
            nn.GELU(approximate='tanh'),
            nn.Linear(dim * mlp_ratio, dim, bias=True),
        )
        self.dropout2 = nn.Dropout(dropout)
        self.adaLN_modulation = nn.Linear(cond_dim, 6 * dim, bias=True)
        self.adaLN_modulation.weight.data.zero_()
        self.adaLN_modulation.bias.data.zero_()
        self.compute_dtype = compute_dtype
        self.seq_axis_attention = seq_axis_attention  # False = reference as written (quirk Q1: batch-axis attention)

    def forward(self, x, rotary_cos_sin, c, other_track=None, other_mask=None, seqlens=None):
        # (seqlens is accepted and unused, as in the reference's cross block, :279-320: its attention is nn.MultiheadAttention)
        cos, sin = rotary_cos_sin
        comp = getattr(cos, "_sedd_compact", None) or Lyr.compact_rotary(cos, sin)
        use_cross = self.enable_cross_attention and other_track is not None
        names = CL.CROSS_BLOCK_PARAMS + (CL.CROSS_ATTN_PARAMS if use_cross else ())
        P = _params(self, names)
        other = other_track.contiguous() if use_cross else None
        return _CrossBlockFn.apply(x, c, other, other_mask if use_cross else None, self, comp[0], comp[1],
                                   torch.is_grad_enabled(), names, *[P[n] for n in names])


class DualTrackDiffusionTransformer(nn.Module):
    """transformers_protlig_cross.py:323-517."""

    def __init__(self, config, vocab_size_protein, vocab_size_ligand, compute_dtype=torch.float32,
                 seq_axis_attention=False):
        super().__init__()
        self.config = config
        self.enable_cross_attention = _cfg(config, "model.enable_cross_attention", True)
        hidden, cond, heads = _cfg(config, "model.hidden_size"), _cfg(config, "model.cond_dim"), _cfg(config, "model.n_heads")
        self.protein_embed = EmbeddingLayer(hidden, vocab_size_protein)
        self.ligand_embed = EmbeddingLayer(hidden, vocab_size_ligand)
        self.sigma_map = TimestepEmbedder(cond)
        self.rotary_emb = Rotary(hidden // heads)
        self.n_blocks = _cfg(config, "model.n_blocks")
        mk = lambda: DDiTBlock(hidden, heads, _cfg(config, "model.dropout"), cond,  # noqa: E731
                               enable_cross_attention=self.enable_cross_attention, compute_dtype=compute_dtype,
                               seq_axis_attention=seq_axis_attention)
        self.blocks_prot = nn.ModuleList([mk() for _ in range(self.n_blocks)])
        self.blocks_lig = nn.ModuleList([mk() for _ in range(self.n_blocks)])
        self.protein_output = DDitFinalLayer(hidden, vocab_size_protein, cond, compute_dtype=compute_dtype)
        self.ligand_output = DDitFinalLayer(hidden, vocab_size_ligand, cond, compute_dtype=compute_dtype)
        self.scale_by_sigma = _cfg(config, "model.scale_by_sigma")
        self.absorb = _cfg(config, "graph.type") == "absorb"

    def _finish(self, x, sigma, indices):
        """:431-446 — sigma rescale (only when scale_by_sigma and absorb) + scatter-zero."""
        return _EpilogueFn.apply(x, sigma, indices, bool(self.scale_by_sigma and self.absorb))

    def forward_single_track(self, indices, sigma, track_type="protein", other_track_hidden=None, other_track_mask=None):
        if track_type == "protein":
            x, output_layer, blocks = self.protein_embed(indices), self.protein_output, self.blocks_prot
        elif track_type == 'ligand':
            x, output_layer, blocks = self.ligand_embed(indices), self.ligand_output, self.blocks_lig
        else:
            raise ValueError('track type not supported')
        c = self.sigma_map(sigma, silu=True)
        rotary_cos_sin = self.rotary_emb(x)
        for i in range(self.n_blocks):
            x = blocks[i](x, rotary_cos_sin, c, other_track=other_track_hidden, other_mask=other_track_mask, seqlens=None)
        return self._finish(output_layer(x, c), sigma, indices)

    def forward_joint(self, protein_indices, ligand_indices, sigma, protein_mask=None, ligand_mask=None):
        protein_hidden = self.protein_embed(protein_indices)
        ligand_hidden = self.ligand_embed(ligand_indices)
        c = self.sigma_map(sigma, silu=True)
        protein_rotary = self.rotary_emb(protein_hidden)
        ligand_rotary = self.rotary_emb(ligand_hidden)
        xc = self.enable_cross_attention
        for i in range(self.n_blocks):
            # both tracks read the OTHER track's pre-update hidden state (:481-501)
            new_protein = self.blocks_prot[i](protein_hidden, protein_rotary, c,
                                              other_track=ligand_hidden if xc else None,
                                              other_mask=ligand_mask if xc else None, seqlens=None)
            new_ligand = self.blocks_lig[i](ligand_hidden, ligand_rotary, c,
                                            other_track=protein_hidden if xc else None,
                                            other_mask=protein_mask if xc else None, seqlens=None)
            protein_hidden, ligand_hidden = new_protein, new_ligand
        protein_logits = self.protein_output(protein_hidden, c)
        ligand_logits = self.ligand_output(ligand_hidden, c)
        return self._finish(protein_logits, sigma, protein_indices), self._finish(ligand_logits, sigma, ligand_indices)


class ProteinLigandDiffusionModel(nn.Module):
    """transformers_protlig_cross.py:520-673."""

    def __init__(self, config, compute_dtype=torch.float32, seq_axis_attention=False):
        super().__init__()
        self.config = config
        self.compute_dtype = compute_dtype
        absorb = _cfg(config, "graph.type") == "absorb"
        vocab_size_protein = 36 + (1 if absorb else 0)
        vocab_size_ligand = 2364 + (1 if absorb else 0)
        self.transformer = DualTrackDiffusionTransformer(config, vocab_size_protein, vocab_size_ligand,
                                                         compute_dtype=compute_dtype,
                                                         seq_axis_attention=seq_axis_attention)
        self._ctx_step, self._ctx_cache, self.context_cache_hits = None, {}, 0

    # ---- per-step context cache for the *_given_* sampling modes -----------------------------------------
    def sampling_step(self, step):
        """Called by sampling.get_pc_sampler's loop: `step` = index of the sampler update about to run (None = sampling
        over).  The conditioning context depends on sigma(t_step), so it is valid for ONE step only: every model call
        of that step with the same conditioning indices (a predictor followed by a corrector, both conditional tracks
        evaluated per step, ...) reuses it instead of re-running the conditioning track's n_blocks // 2 blocks; a new
        step (or None) drops it."""
        if step != self._ctx_step:
            self._ctx_step, self._ctx_cache = step, {}

    def _context(self, indices, timesteps, track):
        """first n_blocks // 2 blocks of the conditioning track without cross-attention (:608-616, :636-642)."""
        key = None
        if self._ctx_step is not None and not torch.is_grad_enabled() and not self.training:
            # (the caller holds `indices` for the whole sampling run, so its storage identifies the conditioning input)
            # and sigma: a second call of the same step at another t (a corrector, a user loop) must not get this context
            key = (track, indices.data_ptr(), indices._version, tuple(indices.shape), timesteps.data_ptr(),
                   timesteps._version, tuple(timesteps.shape))
            hit = self._ctx_cache.get(key)
            if hit is not None:
                self.context_cache_hits += 1
                return hit
        h = self._context_uncached(indices, timesteps, track)
        if key is not None:
            self._ctx_cache[key] = h
        return h

    def _context_uncached(self, indices, timesteps, track):
        tr = self.transformer
        embed, blocks = (tr.protein_embed, tr.blocks_prot) if track == "protein" else (tr.ligand_embed, tr.blocks_lig)
        h = embed(indices)
        c = tr.sigma_map(timesteps, silu=True)
        rotary_cos_sin = tr.rotary_emb(h)
        for i in range(tr.n_blocks // 2):
            h = blocks[i](h, rotary_cos_sin, c, None, None, None)
        return h

    def forward(self, protein_indices=None, ligand_indices=None, timesteps=None, protein_mask=None, ligand_mask=None,
                mode="joint", protein_seq_str=None, ligand_smiles_str=None, use_pretrained_conditioning=False):
        tr = self.transformer
        if use_pretrained_conditioning and (protein_seq_str is not None or ligand_smiles_str is not None):
            raise NotImplementedError("pretrained ESM / MolFormer conditioning needs network-fetched encoders (out of scope)")
        if mode == "protein_only":
            return tr.forward_single_track(protein_indices, timesteps, track_type="protein")
        elif mode == "ligand_only":
            return tr.forward_single_track(ligand_indices, timesteps, track_type="ligand")
        elif mode == "ligand_given_protein":
            protein_hidden = self._context(protein_indices, timesteps, "protein")
            return tr.forward_single_track(ligand_indices, timesteps, track_type="ligand",
                                           other_track_hidden=protein_hidden, other_track_mask=protein_mask)
        elif mode == "protein_given_ligand":
            ligand_hidden = self._context(ligand_indices, timesteps, "ligand")
            return tr.forward_single_track(protein_indices, timesteps, track_type="protein",
                                           other_track_hidden=ligand_hidden, other_track_mask=ligand_mask)
        elif mode == "joint":
            return tr.forward_joint(protein_indices, ligand_indices, timesteps, protein_mask, ligand_mask)
        raise ValueError(f"Unknown mode: {mode}")

    def disable_cross_attention(self):
        for block in list(self.transformer.blocks_prot) + list(self.transformer.blocks_lig):
            block.enable_cross_attention = False

    def enable_cross_attention(self):
        for block in list(self.transformer.blocks_prot) + list(self.transformer.blocks_lig):
            block.enable_cross_attention = True
