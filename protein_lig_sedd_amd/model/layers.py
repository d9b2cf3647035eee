"""Kernel-level building blocks shared by the SEDD and the dual-track cross-attention models: compute-dtype weight
caches, Linear forward/backward on the MFMA GEMM, and the hand-written forward+backward of a DDiT block.

Every tensor operation is a HIP kernel reached through ops.py (C ABI); Python only sequences launches and owns the
saved-activation bookkeeping that torch.autograd would otherwise do.  Gradient convention: if a parameter already has
a `.grad` tensor (the trainer pre-allocates them as views of one flat buffer) weight gradients are ACCUMULATED into
it in place by the producing kernel and autograd receives None; otherwise a fresh gradient tensor is returned.
"""
import os

import torch

from .. import ops



def compute_weight(W, dtype, transposed=False):
    """W (fp32 master parameter [N, K]) in the compute dtype, optionally transposed ([K, N]); cached per parameter
    version so the cast runs once per optimiser step (what autocast's weight cache does in the reference)."""
    ent = getattr(W, "_sedd_wc", None)
    tag = (W._version, W.data_ptr())
    if ent is None or ent["tag"] != tag:
        ent = {"tag": tag}
        W._sedd_wc = ent
    key = (dtype, transposed)
    t = ent.get(key)
    if t is None:
        Wd = W.detach()
        if transposed:  # [K, Np]: the out-feature dim (reduction dim of the dgrad GEMM) zero-padded to the MFMA K step
            t = transpose_padded(Wd, 8 if dtype == torch.bfloat16 else 4, dtype)
        else:
            t = Wd if dtype == torch.float32 else ops.cast(Wd, dtype)
        ent[key] = t
    return t


def set_compute_weight(W, dtype, tensor):
    """Let the optimiser hand over a compute-dtype copy it produced itself (fused AdamW -> bf16 write)."""
    W._sedd_wc = {"tag": (W._version, W.data_ptr()), (dtype, False): tensor}


def transpose_padded(a, align, out_dtype=None):
    """a [M, N] -> [N, Mp] with Mp = M rounded up to `align` and a zero tail (the reduction dim of a wgrad GEMM)."""
    M, N = a.shape
    Mp = (M + align - 1) // align * align
    if Mp == M:
        return ops.transpose(a, out_dtype)
    out = torch.empty(N, Mp, dtype=out_dtype or a.dtype, device=a.device)
    out[:, M:].zero_()
    ops.check(ops.lib().sedd_transpose(a.data_ptr(), ops._dt(a), a.stride(0), out.data_ptr(), ops._dt(out), Mp, M, N,
                                       ops._stream()), "sedd_transpose")
    return out


TN_COLSUM = not os.environ.get("SEDD_NO_TN_COLSUM")  # A/B switch: bias gradients out of the weight-gradient launch


_CAST_ONCE = [None, None, None]


def cast_once(t, dtype):
    """the conditioning vector c in the compute dtype: every block of a forward pass casts the SAME tensor (24 blocks of the
    dual-track model, 12 + the final layer of SEDD) — one launch, remembered for as long as the next request is for the same
    storage at the same version on the same stream.  The source tensor is held too: while it is, its memory cannot be handed
    to another tensor, so (address, version) does identify the contents."""
    key = (t.data_ptr(), t._version, tuple(t.shape), dtype, ops._stream())
    if _CAST_ONCE[0] != key:
        _CAST_ONCE[:] = [key, t, ops.cast(t, dtype)]
    return _CAST_ONCE[2]


def _sink(param):
    """(grad tensor to accumulate into or None, whether autograd should receive the result)."""
    return param.grad if (param is not None and param.grad is not None) else None


def linear_fwd(x2d, W, bias=None, gelu=False, want_pre=False, out_dtype=None):
    """nn.Linear forward: x2d [M, K] (compute dtype) @ W[N, K]^T."""
    return ops.gemm_nt(x2d, compute_weight(W, x2d.dtype), bias=None if bias is None else bias.detach(), gelu=gelu,
                       want_pre=want_pre, out_dtype=out_dtype)


def _weight_grads(x2d, dy2d, W, bias):
    """dW (+ db) of a Linear into the parameters' gradient sinks, or as fresh tensors: -> (dW or None, db or None)."""
    align = 8 if dy2d.dtype == torch.bfloat16 else 4
    sink = _sink(W)
    if ops.gemm_tn_supported(dy2d, x2d):  # dW = dY^T X straight from the row-major activations
        if sink is not None:
            bs = _sink(bias)
            if bs is not None and dy2d.dtype == torch.bfloat16 and bs.is_contiguous() and TN_COLSUM:
                ops.gemm_tn(dy2d, x2d, out=sink, accumulate=True, colsum=bs)  # db out of the same launch
                return None, None
            ops.gemm_tn(dy2d, x2d, out=sink, accumulate=True)
            dW = None
        else:
            dW = ops.gemm_tn(dy2d, x2d, out_dtype=torch.float32)
    else:
        dyT = transpose_padded(dy2d, align)
        xT = transpose_padded(x2d, align)
        if sink is not None:
            ops.gemm_nt(dyT, xT, out=sink, accumulate=True)
            dW = None
        else:
            dW = ops.gemm_nt(dyT, xT, out_dtype=torch.float32)
    db = None
    if bias is not None:
        bs = _sink(bias)
        if bs is not None:
            ops.colsum(dy2d, out=bs, accumulate=True)
        else:
            db = ops.colsum(dy2d)
    return dW, db


def linear_bwd(x2d, dy2d, W, bias=None, need_dx=True, dx_dtype=None):
    """-> (dx [M, K] or None, dW for autograd or None, db for autograd or None)."""
    align = 8 if dy2d.dtype == torch.bfloat16 else 4
    side = ops.SIDE
    if side is not None and _sink(W) is not None and (bias is None or _sink(bias) is not None):
        # the weight / bias gradients land in the flat gradient buffer, which only the optimiser reads: off the critical
        # path of backward, so they go to the side stream (ops.SideStream) and the data gradient carries on here
        side.defer(lambda x=x2d, dy=dy2d: _weight_grads(x, dy, W, bias), x2d, dy2d,
                   torch_ops=not ops.gemm_tn_supported(dy2d, x2d))  # (bound now: dy2d may be re-bound to a padded copy below)
        dW = db = None
    else:
        dW, db = _weight_grads(x2d, dy2d, W, bias)
    dx = None
    if need_dx:
        Wc = compute_weight(W, dy2d.dtype)                        # [N, K] as stored
        if dx_dtype in (None, dy2d.dtype) and ops.gemm_nn_supported(dy2d, Wc):
            return ops.gemm_nn(dy2d, Wc), dW, db                  # dX = dY W, no transposed weight copy
        Wt = compute_weight(W, dy2d.dtype, transposed=True)      # [K, Np]
        if Wt.shape[1] != dy2d.shape[1] or dy2d.stride(0) % align:  # ragged out-feature dim (e.g. V = 26): pad dy
            dyp = torch.zeros(dy2d.shape[0], Wt.shape[1], dtype=dy2d.dtype, device=dy2d.device)
            dyp[:, :dy2d.shape[1]].copy_(dy2d)
            dy2d = dyp
        if (dx_dtype == torch.float32 and dy2d.dtype == torch.bfloat16 and dy2d.shape[0] <= 128 and Wt.shape[1] >= 1024
                and not os.environ.get("SEDD_NO_SMALL_SPLITK")):
            # few rows, long reduction (adaLN: dc [B, C] = dmod [B, 6D] W): one output tile, so let the kernel split K
            # over blocks — which it does for accumulating fp32 outputs — instead of one block walking all of K
            dx = torch.zeros(dy2d.shape[0], Wt.shape[0], dtype=torch.float32, device=dy2d.device)
            ops.gemm_nt(dy2d, Wt, out=dx, accumulate=True)
        else:
            dx = ops.gemm_nt(dy2d, Wt, out_dtype=dx_dtype)
    return dx, dW, db


def mlp_out_bwd(g2d, u2d, dy2d, W2, b2, b1, u_is_grad=False):
    """backward of `y = Linear2(GELU(u))` down to du: -> (du, dW2, db2, db1).  The data gradient, GELU's derivative and
    the first Linear's bias gradient (column sums of du) are one GEMM launch when the shapes allow.  u_is_grad: `u2d`
    holds gelu'(u) (the forward's GELU_GRAD epilogue) and the launch's epilogue is one multiply."""
    W2c = compute_weight(W2, dy2d.dtype)
    if u_is_grad:
        if not (ops.gemm_nn_supported(dy2d, W2c) and u2d.is_contiguous() and u2d.data_ptr() % 16 == 0):
            raise RuntimeError("mlp_out_bwd: the forward stored gelu' for a shape the fused data gradient cannot take")
        _, dW2, db2 = linear_bwd(g2d, dy2d, W2, b2, need_dx=False)
        bs = _sink(b1)
        db1 = None
        if bs is None:
            bs = db1 = torch.zeros(u2d.shape[1], dtype=torch.float32, device=u2d.device)
        return ops.gemm_dmul(dy2d, W2c, u2d, colsum=bs), dW2, db2, db1
    if (ops.gemm_nn_supported(dy2d, W2c) and u2d.dtype == torch.bfloat16 and u2d.is_contiguous()
            and u2d.data_ptr() % 16 == 0 and not os.environ.get("SEDD_NO_DGELU")):
        _, dW2, db2 = linear_bwd(g2d, dy2d, W2, b2, need_dx=False)
        bs = _sink(b1)
        db1 = None
        if bs is None:
            bs = db1 = torch.zeros(u2d.shape[1], dtype=torch.float32, device=u2d.device)
        du = ops.gemm_nn_dgelu(dy2d, W2c, u2d, colsum=bs)
        return du, dW2, db2, db1
    W2t = compute_weight(W2, dy2d.dtype, transposed=True)
    if W2t.shape[1] == dy2d.shape[1] and ops.gemm_nt_dgelu_supported(dy2d, W2t, u2d):
        _, dW2, db2 = linear_bwd(g2d, dy2d, W2, b2, need_dx=False)
        bs = _sink(b1)
        db1 = None
        if bs is None:
            bs = db1 = torch.zeros(u2d.shape[1], dtype=torch.float32, device=u2d.device)
        du = ops.gemm_nt_dgelu(dy2d, W2t, u2d, colsum=bs)
        return du, dW2, db2, db1
    dg, dW2, db2 = linear_bwd(g2d, dy2d, W2, b2)
    du = ops.gelu_bwd(u2d, dg)
    bs = _sink(b1)
    if bs is not None:
        ops.colsum(du, out=bs, accumulate=True)
        return du, dW2, db2, None
    return du, dW2, db2, ops.colsum(du)


def _vec_sink(param, D):
    s = _sink(param)
    if s is not None:
        return s, None
    z = torch.zeros(D, dtype=torch.float32, device=param.device)
    return z, z


def compact_rotary(cos, sin):
    """reference tables [1, L, 3, 1, d] (rotary.py:17-25) -> contiguous fp32 [L, d/2] halves for the kernel."""
    d = cos.shape[-1]
    return (cos[0, :, 0, 0, :d // 2].float().contiguous(), sin[0, :, 0, 0, :d // 2].float().contiguous())


def dropout_mask(shape, p, device):
    """dropout state of one bias_dropout_add_scale site: a (seed, p) pair — the kernels derive the keep decision of
    every element from a counter-based hash, forward and backward alike (no [M, D] mask tensor, no torch RNG kernel)."""
    return ops.DropSeed(p)


# ------------------------------------------------------------------------------------------------- DDiT block
class BlockLink:
    """What two CONSECUTIVE DDiT blocks share so that the passes over the residual stream at their boundary fuse
    (transformer.py:188-189 of block i feeding :165-166 of block i+1): forward — block i's last gated residual and block
    i+1's first LayerNorm-modulate are one kernel (sedd_gate_residual_ln_modulate_fwd: x2 is written once and not re-read),
    for which block i evaluates block i+1's adaLN projection and hands over (mod, h1, mean1, rstd1); backward — block i+1's
    norm1 backward and block i's gated-residual backward (with mlp[2]'s bias gradient) are one kernel
    (sedd_ln_modulate_bwd_gate_colsum) and block i+1 hands back (dy, dmod with the gate slice filled, a fresh bias gradient
    when the parameter has no gradient sink).  Built by _SEDD._forward only: there x2 has exactly one consumer, which is
    what lets block i+1's dx stand in for autograd's gradient of x2.  A block that finds its link empty runs stand-alone.
    Guards: the next block takes the hand-over only if its input IS the x2 the link was filled for (a forward hook that
    replaces a block's output breaks the chain silently and correctly); in backward, the gradient autograd delivers for x2
    must be the very dx the next block returned — a second consumer of x2 (an activation hook feeding the loss) makes
    autograd sum into another tensor, which is detected and refused (SEDD_NO_BLOCK_CHAIN=1 then)."""
    __slots__ = ("norm1_w", "ada_w", "ada_b", "mlp2_bias", "mod", "h1", "mean1", "rstd1", "prev", "dy", "dmod_prev",
                 "db2_fresh", "x2_key", "dx_key")

    def __init__(self, prev_block, next_block):
        # next_block: a DDiTBlock, or the DDitFinalLayer behind the last block (its LayerNorm is `norm_final` and its
        # modulation [B, 2D]; the block lays that into a [B, 6D]-strided buffer so that one row stride serves the kernel)
        norm = next_block.norm1 if hasattr(next_block, "norm1") else next_block.norm_final
        self.norm1_w, self.ada_w, self.ada_b = norm.weight, next_block.adaLN_modulation.weight, next_block.adaLN_modulation.bias
        self.mlp2_bias = prev_block.mlp[2].bias
        self.x2_key = self.dx_key = None
        self.mod = self.h1 = self.mean1 = self.rstd1 = self.prev = self.dy = self.dmod_prev = self.db2_fresh = None

    @staticmethod
    def key(t):
        return (t.data_ptr(), t._version, tuple(t.shape))

    def taken_by(self, x):
        """whether the hand-over was made for exactly this input tensor (else it is dropped)."""
        if self.h1 is None:
            return False
        if self.x2_key != BlockLink.key(x):
            self.mod = self.h1 = self.mean1 = self.rstd1 = self.prev = None
            return False
        return True

    def check_grad(self, dx2):
        if self.dx_key != BlockLink.key(dx2):
            raise RuntimeError("block-boundary fusion: the output of a DDiT block has a second consumer (the gradient "
                               "autograd delivers is not the next block's dx); run with SEDD_NO_BLOCK_CHAIN=1")


BLOCK_CHAIN = not os.environ.get("SEDD_NO_BLOCK_CHAIN")  # A/B switch for the boundary fusion


BLOCK_PARAMS = ("norm1.weight", "attn_qkv.weight", "attn_out.weight", "norm2.weight", "mlp.0.weight", "mlp.0.bias",
                "mlp.2.weight", "mlp.2.bias", "adaLN_modulation.weight", "adaLN_modulation.bias")


def ddit_block_forward(x, c, P, n_heads, cos, sin, p_drop, cdt, save, cu_seqlens=None, link_in=None, link_out=None):
    """DDiTBlock.forward (model/transformer.py:153-189).  x fp32 [B, L, D]; c fp32 [B, C]; P: dict of the block's
    parameters by reference name; cos/sin compact rotary tables.  Returns x_out fp32 and (if save) the saved state.
    link_in / link_out: BlockLink with the previous / next block (None: stand-alone)."""
    B, L, D = x.shape
    M, H, d = B * L, n_heads, D // n_heads
    chained_in = link_in is not None and link_in.taken_by(x)
    c_c = None
    w1, w2 = P["norm1.weight"].detach(), P["norm2.weight"].detach()
    if chained_in:  # the previous block's last pass already produced this block's modulation and LayerNorm-modulate
        mod, h1, mean1, rstd1 = link_in.mod, link_in.h1, link_in.mean1, link_in.rstd1
        link_in.mod = link_in.h1 = link_in.mean1 = link_in.rstd1 = None
    else:
        c_c = c if cdt == torch.float32 else cast_once(c, cdt)
        mod = linear_fwd(c_c, P["adaLN_modulation.weight"], P["adaLN_modulation.bias"])            # [B, 6D]
        h1, mean1, rstd1 = ops.ln_modulate_fwd(x, w1, shift=mod[:, :D], scale=mod[:, D:2 * D], out_dtype=cdt)
    Wqkv = compute_weight(P["attn_qkv.weight"], cdt)
    if ops.gemm_nt_rotary_supported(h1.view(M, D), Wqkv, L, H, d):  # rotary rides in the projection's epilogue
        qkv = ops.gemm_nt_rotary(h1.view(M, D), Wqkv, cos, sin, L, H, d).view(B, L, 3, H, d)
    else:
        qkv = linear_fwd(h1.view(M, D), P["attn_qkv.weight"]).view(B, L, 3, H, d)
        ops.rotary_qkv_(qkv, cos, sin)
    if cu_seqlens is None:
        o, lse = ops.attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
    else:  # variable-length packing: attention inside the segments of the flattened token axis (transformer.py:173-181)
        o, lse = ops.attn_varlen_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], cu_seqlens, L)
    a = linear_fwd(o.view(M, D), P["attn_out.weight"]).view(B, L, D)
    inv = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
    keep1 = dropout_mask((B, L, D), p_drop, x.device) if p_drop > 0 else None
    if ops.gate_residual_ln_modulate_supported(x, a, keep1):  # x1 and LN(x1) in one pass over the rows
        x1, h2, mean2, rstd2 = ops.gate_residual_ln_modulate_fwd(x, a, mod[:, 2 * D:3 * D], keep1, inv, w2,
                                                                 mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D])
    else:
        x1 = ops.gate_residual_fwd(x, a, mod[:, 2 * D:3 * D], keep1, inv)
        h2, mean2, rstd2 = ops.ln_modulate_fwd(x1, w2, shift=mod[:, 3 * D:4 * D], scale=mod[:, 4 * D:5 * D], out_dtype=cdt)
    W1 = P["mlp.0.weight"]
    # the backward's fused data gradient through GELU multiplies by gelu'(u): when it will run (static shape test) the
    # forward stores that derivative — from the sigmoid it evaluates anyway — in place of u
    u_grad = save and ops.gemm_dmul_shapes_ok(M, W1.shape[0], D, cdt) and P["mlp.2.weight"].stride(0) % 8 == 0
    if save:
        g, u = linear_fwd(h2.view(M, D), W1, P["mlp.0.bias"], gelu=True, want_pre="grad" if u_grad else True)
    else:
        g, u = linear_fwd(h2.view(M, D), P["mlp.0.weight"], P["mlp.0.bias"], gelu=True), None
    y = linear_fwd(g, P["mlp.2.weight"], P["mlp.2.bias"]).view(B, L, D)
    keep2 = dropout_mask((B, L, D), p_drop, x.device) if p_drop > 0 else None
    chained_out = link_out is not None and ops.gate_residual_ln_modulate_supported(x1, y, keep2)
    if chained_out:  # x2 and the NEXT block's LayerNorm-modulate of it in one pass over the rows
        if c_c is None:
            c_c = c if cdt == torch.float32 else cast_once(c, cdt)
        if link_out.ada_w.shape[0] == 6 * D:
            mod_n = linear_fwd(c_c, link_out.ada_w, link_out.ada_b)
        else:  # the final layer's [B, 2D] modulation, at this block's row stride
            mod_n = torch.empty(B, 6 * D, dtype=cdt, device=x.device)[:, :link_out.ada_w.shape[0]]
            ops.gemm_nt(c_c, compute_weight(link_out.ada_w, cdt), bias=link_out.ada_b.detach(), out=mod_n)
        x2, link_out.h1, link_out.mean1, link_out.rstd1 = ops.gate_residual_ln_modulate_fwd(
            x1, y, mod[:, 5 * D:6 * D], keep2, inv, link_out.norm1_w.detach(), mod_n[:, :D], mod_n[:, D:2 * D])
        link_out.mod, link_out.x2_key = mod_n, BlockLink.key(x2)
        link_out.prev = dict(y=y, gate=mod[:, 5 * D:6 * D], keep2=keep2, inv=inv) if save else None
    else:
        x2 = ops.gate_residual_fwd(x1, y, mod[:, 5 * D:6 * D], keep2, inv)
    st = None
    if save:
        st = dict(x=x, c=c, mod=mod, h1=h1, mean1=mean1, rstd1=rstd1, qkv=qkv, o=o, lse=lse, a=a, keep1=keep1,
                  x1=x1, h2=h2, mean2=mean2, rstd2=rstd2, u=u, u_grad=u_grad, g=g, y=y, keep2=keep2, inv=inv, cos=cos,
                  sin=sin, cu=cu_seqlens, link_in=link_in if chained_in else None,
                  link_out=link_out if chained_out else None)
    return x2, st


def ddit_block_backward(dx2, st, P, n_heads):
    """-> (dx fp32 [B,L,D], dc fp32 [B,C], {param name: grad or None})"""
    x = st["x"]
    B, L, D = x.shape
    M, H, d = B * L, n_heads, D // n_heads
    mod, cdt = st["mod"], st["h1"].dtype
    grads = {}
    dx = torch.empty_like(dx2)  # = dx2 + both LN-branch gradients (dx2 itself is autograd's and stays untouched)
    # ---- MLP branch: x2 = x1 + gate_mlp * dropout(mlp(modulate(LN(x1))))
    lo = st.get("link_out")
    if lo is not None and lo.dy is not None:
        # the next block's norm1 backward already sent its finished dx (= dx2) through this gated residual's backward
        lo.check_grad(dx2)
        dy, dmod, grads["mlp.2.bias"] = lo.dy.view(M, D), lo.dmod_prev, lo.db2_fresh
        lo.dy = lo.dmod_prev = lo.db2_fresh = lo.prev = None
    else:
        dmod = torch.empty(B, 6 * D, dtype=torch.float32, device=x.device)
        db2, grads["mlp.2.bias"] = _vec_sink(P["mlp.2.bias"], D)  # column sums of dy ride along in the same pass
        dy = ops.gate_residual_bwd(dx2, st["y"], mod[:, 5 * D:6 * D], dmod[:, 5 * D:6 * D], st["keep2"], st["inv"],
                                   dbias=db2).view(M, D)
    du, grads["mlp.2.weight"], _, grads["mlp.0.bias"] = mlp_out_bwd(
        st["g"], st["u"], dy, P["mlp.2.weight"], None, P["mlp.0.bias"], u_is_grad=st["u_grad"])
    dh2, grads["mlp.0.weight"], _ = linear_bwd(st["h2"].view(M, D), du, P["mlp.0.weight"], None)
    dw2, grads["norm2.weight"] = _vec_sink(P["norm2.weight"], D)
    # ---- attention branch: x1 = x + gate_msa * dropout(attn_out(attn(rotary(qkv(modulate(LN(x)))))))
    if ops.gate_residual_ln_modulate_supported(st["x1"], st["a"], st["keep1"]) and dh2.dtype == st["a"].dtype:
        # norm2's backward and the gated residual's backward share one pass over the rows of dx
        da = ops.ln_modulate_bwd_gate(dh2.view(B, L, D), st["x1"], P["norm2.weight"].detach(), mod[:, 4 * D:5 * D],
                                      st["mean2"], st["rstd2"], dx, dx2, dw2, dmod[:, 3 * D:4 * D], dmod[:, 4 * D:5 * D],
                                      st["a"], mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], st["keep1"], st["inv"]).view(M, D)
    else:
        ops.ln_modulate_bwd(dh2.view(B, L, D), st["x1"], P["norm2.weight"].detach(), mod[:, 4 * D:5 * D], st["mean2"],
                            st["rstd2"], dx, dx2, dw2, dmod[:, 3 * D:4 * D], dmod[:, 4 * D:5 * D])
        da = ops.gate_residual_bwd(dx, st["a"], mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], st["keep1"], st["inv"]).view(M, D)
    do, grads["attn_out.weight"], _ = linear_bwd(st["o"].view(M, D), da, P["attn_out.weight"])
    qkv = st["qkv"]
    dqkv = torch.empty_like(qkv)
    if st.get("cu") is None:
        ops.attn_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], st["o"], do.view(B, L, H, d), st["lse"], dqkv[:, :, 0],
                     dqkv[:, :, 1], dqkv[:, :, 2], rotary=(st["cos"], st["sin"]))  # rotary's gradient fused into the stores
    else:
        dqkv.zero_()  # (tokens outside every segment receive no gradient)
        ops.attn_varlen_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], st["o"], do.view(B, L, H, d), st["lse"], dqkv[:, :, 0],
                            dqkv[:, :, 1], dqkv[:, :, 2], st["cu"], L)
        ops.rotary_qkv_(dqkv, st["cos"], st["sin"], conj=True)  # positions are row positions: not fused into the stores
    dh1, grads["attn_qkv.weight"], _ = linear_bwd(st["h1"].view(M, D), dqkv.view(M, 3 * D), P["attn_qkv.weight"])
    dw1, grads["norm1.weight"] = _vec_sink(P["norm1.weight"], D)
    li = st.get("link_in")
    if li is not None and li.prev is not None and dh1.dtype == li.prev["y"].dtype:
        # block boundary: norm1's backward and the PREVIOUS block's last gated residual's backward (with its mlp[2] bias
        # gradient) share one pass over the rows of the finished dx
        pv = li.prev
        li.dmod_prev = torch.empty(B, 6 * D, dtype=torch.float32, device=x.device)
        db2p, li.db2_fresh = _vec_sink(li.mlp2_bias, D)
        li.dy = ops.ln_modulate_bwd_gate(dh1.view(B, L, D), x, P["norm1.weight"].detach(), mod[:, D:2 * D], st["mean1"],
                                         st["rstd1"], dx, True, dw1, dmod[:, :D], dmod[:, D:2 * D], pv["y"], pv["gate"],
                                         li.dmod_prev[:, 5 * D:6 * D], pv["keep2"], pv["inv"], dbias=db2p)
        li.dx_key = BlockLink.key(dx)
    else:
        ops.ln_modulate_bwd(dh1.view(B, L, D), x, P["norm1.weight"].detach(), mod[:, D:2 * D], st["mean1"], st["rstd1"],
                            dx, True, dw1, dmod[:, :D], dmod[:, D:2 * D])
    # ---- adaLN: mod = Linear(c)
    # fp32 operands whatever the compute dtype: dmod [B, 6D] and c [B, C] are fp32 as they stand and a batch is a few dozen
    # rows, so the exact-fp32 kernels take them as stored (weight gradient from [B, .] row-major, dc against the master
    # weight with K split over blocks) — the bf16 route needed a cast, a transposed weight copy and a zero-fill per block
    dc, grads["adaLN_modulation.weight"], grads["adaLN_modulation.bias"] = linear_bwd(
        st["c"], dmod, P["adaLN_modulation.weight"], P["adaLN_modulation.bias"])
    return dx, dc, grads


# ------------------------------------------------------------------------------------------------ final layer
FINAL_PARAMS = ("norm_final.weight", "linear.weight", "linear.bias", "adaLN_modulation.weight", "adaLN_modulation.bias")


def final_layer_forward(x, c, P, cdt, save, link_in=None):
    """DDitFinalLayer.forward (model/transformer.py:234-238) -> logits [B, L, V] in the compute dtype.
    link_in: BlockLink with the last block (the LayerNorm-modulate then came out of that block's last pass)."""
    B, L, D = x.shape
    M = B * L
    chained = link_in is not None and link_in.taken_by(x)
    if chained:
        mod, h, mean, rstd = link_in.mod, link_in.h1, link_in.mean1, link_in.rstd1
        link_in.mod = link_in.h1 = link_in.mean1 = link_in.rstd1 = None
    else:
        c_c = c if cdt == torch.float32 else cast_once(c, cdt)
        mod = linear_fwd(c_c, P["adaLN_modulation.weight"], P["adaLN_modulation.bias"])            # [B, 2D]
        h, mean, rstd = ops.ln_modulate_fwd(x, P["norm_final.weight"].detach(), shift=mod[:, :D], scale=mod[:, D:2 * D],
                                            out_dtype=cdt)
    logits = linear_fwd(h.view(M, D), P["linear.weight"], P["linear.bias"]).view(B, L, -1)
    st = dict(x=x, c=c, mod=mod, h=h, mean=mean, rstd=rstd, link_in=link_in if chained else None) if save else None
    return logits, st


def final_layer_backward(dlogits, st, P):
    x = st["x"]
    B, L, D = x.shape
    M = B * L
    mod, cdt = st["mod"], st["h"].dtype
    grads = {}
    dl = dlogits.reshape(M, -1)
    if dl.dtype != cdt:
        dl = ops.cast(dl, cdt)
    dh, grads["linear.weight"], grads["linear.bias"] = linear_bwd(st["h"].view(M, D), dl, P["linear.weight"], P["linear.bias"])
    dmod = torch.empty(B, 2 * D, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dw, grads["norm_final.weight"] = _vec_sink(P["norm_final.weight"], D)
    li = st.get("link_in")
    if li is not None and li.prev is not None and dh.dtype == li.prev["y"].dtype:  # block boundary: see ddit_block_backward
        pv = li.prev
        li.dmod_prev = torch.empty(B, 6 * D, dtype=torch.float32, device=x.device)
        db2p, li.db2_fresh = _vec_sink(li.mlp2_bias, D)
        li.dy = ops.ln_modulate_bwd_gate(dh.view(B, L, D), x, P["norm_final.weight"].detach(), mod[:, D:2 * D], st["mean"],
                                         st["rstd"], dx, False, dw, dmod[:, :D], dmod[:, D:2 * D], pv["y"], pv["gate"],
                                         li.dmod_prev[:, 5 * D:6 * D], pv["keep2"], pv["inv"], dbias=db2p)
        li.dx_key = BlockLink.key(dx)
    else:
        ops.ln_modulate_bwd(dh.view(B, L, D), x, P["norm_final.weight"].detach(), mod[:, D:2 * D], st["mean"], st["rstd"],
                            dx, False, dw, dmod[:, :D], dmod[:, D:2 * D])
    # fp32 operands whatever the compute dtype: dmod [B, 6D] and c [B, C] are fp32 as they stand and a batch is a few dozen
    # rows, so the exact-fp32 kernels take them as stored (weight gradient from [B, .] row-major, dc against the master
    # weight with K split over blocks) — the bf16 route needed a cast, a transposed weight copy and a zero-fill per block
    dc, grads["adaLN_modulation.weight"], grads["adaLN_modulation.bias"] = linear_bwd(
        st["c"], dmod, P["adaLN_modulation.weight"], P["adaLN_modulation.bias"])
    return dx, dc, grads


# ------------------------------------------------------------------------------------------- sigma embedding
SIGMA_PARAMS = ("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")


def sigma_map_forward(sigma, P, freq_dim, outer_silu, save):
    """TimestepEmbedder.forward (transformer.py:91-94), optionally fused with the caller's F.silu (:285). fp32."""
    emb = ops.timestep_embedding(sigma, freq_dim)
    h0 = linear_fwd(emb, P["mlp.0.weight"], P["mlp.0.bias"])
    s0 = ops.silu_fwd(h0)
    h1 = linear_fwd(s0, P["mlp.2.weight"], P["mlp.2.bias"])
    out = ops.silu_fwd(h1) if outer_silu else h1
    st = dict(emb=emb, h0=h0, s0=s0, h1=h1, outer=outer_silu) if save else None
    return out, st


def sigma_map_backward(dout, st, P):
    grads = {}
    dh1 = ops.silu_bwd(st["h1"], dout) if st["outer"] else dout.contiguous()
    ds0, grads["mlp.2.weight"], grads["mlp.2.bias"] = linear_bwd(st["s0"], dh1, P["mlp.2.weight"], P["mlp.2.bias"])
    dh0 = ops.silu_bwd(st["h0"], ds0)
    _, grads["mlp.0.weight"], grads["mlp.0.bias"] = linear_bwd(st["emb"], dh0, P["mlp.0.weight"], P["mlp.0.bias"],
                                                               need_dx=False)
    return grads
