"""Hand-written forward + backward of the dual-track DDiT block with cross-attention
(protlig_dd/model/transformers_protlig_cross.py:237-320 `DDiTBlock`, :107-140 `CrossAttentionLayer`) on the HIP
kernels.  Reproduces the reference AS WRITTEN:

  * quirk Q1: the "self-attention" nn.MultiheadAttention is built with batch_first=True but is fed
    [L, B, D] tensors (:301-305), so it attends over the BATCH axis for every sequence position.  Here that is the same
    strided attention kernel called with (batch stride, sequence stride) swapped — no transposes are materialised.
    `seq_axis_attention=True` selects the intended sequence-axis attention instead.
  * nn.MultiheadAttention re-projects q, k, v with its own in_proj_weight / in_proj_bias and out_proj (:251, :113).
  * cross-attention reads the RAW hidden state of the other track as keys/values, with an optional key mask.

Not reproduced: the dropout nn.MultiheadAttention applies to the attention probabilities in train mode (its RNG stream
cannot match torch's anyway); the two residual dropouts and CrossAttentionLayer.dropout are applied (uint8 keep masks).
"""
import torch

from .. import ops
from . import layers as Lyr
from .layers import _sink, compute_weight, transpose_padded

CROSS_BLOCK_PARAMS = ("norm1.weight", "attn_qkv.weight", "attn_qkv.bias", "attn.in_proj_weight", "attn.in_proj_bias",
                      "attn.out_proj.weight", "attn.out_proj.bias", "attn_out.weight", "attn_out.bias", "norm2.weight",
                      "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias", "adaLN_modulation.weight",
                      "adaLN_modulation.bias")
CROSS_ATTN_PARAMS = ("cross_attn.cross_attn.in_proj_weight", "cross_attn.cross_attn.in_proj_bias",
                     "cross_attn.cross_attn.out_proj.weight", "cross_attn.cross_attn.out_proj.bias",
                     "cross_attn.norm.weight", "cross_attn.norm.bias")


class GradSink:
    """Per-parameter gradient destination: the parameter's existing .grad (trainer's flat buffer) or a fresh zero tensor
    that is handed back to autograd.  Lets several kernels accumulate into row slices of one packed parameter."""

    def __init__(self):
        self.fresh = {}

    def get(self, name, param):
        if param.grad is not None:
            return param.grad
        if name not in self.fresh:
            self.fresh[name] = torch.zeros_like(param, dtype=torch.float32)
        return self.fresh[name]

    def result(self, names):
        return {n: self.fresh.get(n) for n in names}


_ONES = {}


def _ones(B, D, device):
    """the all-ones [B, D] gate of the cross-attention residual (x + dropout(x_ca)): read-only, one per shape and device
    instead of a fill launch per block and step."""
    key = (B, D, device)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(B, D, dtype=torch.float32, device=device)
    return t


def part_fwd(x2d, W, b, r0, r1, out_dtype=None):
    """rows [r0, r1) of a packed projection: x2d @ W[r0:r1]^T + b[r0:r1]"""
    return ops.gemm_nt(x2d, compute_weight(W, x2d.dtype)[r0:r1], bias=None if b is None else b.detach()[r0:r1],
                       out_dtype=out_dtype)


def part_bwd(x2d, dy2d, W, b, r0, r1, gW, gb, need_dx=True, dx_dtype=None):
    """accumulate dW[r0:r1] += dy^T x, db[r0:r1] += colsum(dy); return dx = dy @ W[r0:r1] (or None)."""
    def wgrads():
        if ops.gemm_tn_supported(dy2d, x2d):  # straight from the row-major activations
            if gb is not None and dy2d.dtype == torch.bfloat16 and Lyr.TN_COLSUM:
                ops.gemm_tn(dy2d, x2d, out=gW[r0:r1], accumulate=True, colsum=gb[r0:r1])  # db out of the same launch
                return
            ops.gemm_tn(dy2d, x2d, out=gW[r0:r1], accumulate=True)
        else:
            align = 8 if dy2d.dtype == torch.bfloat16 else 4
            ops.gemm_nt(transpose_padded(dy2d, align), transpose_padded(x2d, align), out=gW[r0:r1], accumulate=True)
        if gb is not None:
            ops.colsum(dy2d, out=gb[r0:r1], accumulate=True)

    side = ops.SIDE
    if side is not None and W.grad is not None and (b is None or gb is None or b.grad is not None):
        # into the flat gradient buffer, read by the optimiser only: see layers.linear_bwd
        side.defer(wgrads, x2d, dy2d, torch_ops=not ops.gemm_tn_supported(dy2d, x2d))
    else:
        wgrads()
    if not need_dx:
        return None
    return part_dgrad(dy2d, W, r0, r1, dx_dtype=dx_dtype)


def part_dgrad(dy2d, W, r0, r1, out=None, dx_dtype=None):
    """dx = dy @ W[r0:r1]: against the weight rows as stored when the NN kernel takes the operands, else against the
    cached transposed copy."""
    Wc = compute_weight(W, dy2d.dtype)[r0:r1]
    if dx_dtype in (None, dy2d.dtype) and (out is None or out.dtype == dy2d.dtype) and ops.gemm_nn_supported(dy2d, Wc):
        return ops.gemm_nn(dy2d, Wc, out=out)
    Wt = compute_weight(W, dy2d.dtype, transposed=True)  # [K, Np]
    if out is not None:
        return ops.gemm_nt(dy2d, Wt[:, r0:r1], out=out)
    return ops.gemm_nt(dy2d, Wt[:, r0:r1], out_dtype=dx_dtype)


def _self_attn_views(t, B, L, H, d, seq_axis):
    """t: [B, L, H, d]-shaped (possibly strided) tensor -> the view the attention kernel should see."""
    return t if seq_axis else t.transpose(0, 1)  # as written: 'batch' = sequence position, 'sequence' = batch index


def cross_block_forward(x, c, other, other_mask, P, n_heads, cos, sin, p_drop, cdt, save, seq_axis=False):
    """DDiTBlock.forward (transformers_protlig_cross.py:279-320). x fp32 [B, L, D]; other: None or [B, Lo, D] fp32."""
    B, L, D = x.shape
    M, H, d = B * L, n_heads, D // n_heads
    c_c = c if cdt == torch.float32 else Lyr.cast_once(c, cdt)
    mod = Lyr.linear_fwd(c_c, P["adaLN_modulation.weight"], P["adaLN_modulation.bias"])
    h1, mean1, rstd1 = ops.ln_modulate_fwd(x, P["norm1.weight"].detach(), shift=mod[:, :D], scale=mod[:, D:2 * D],
                                           out_dtype=cdt)
    qkv = Lyr.linear_fwd(h1.view(M, D), P["attn_qkv.weight"], P["attn_qkv.bias"]).view(B, L, 3, H, d)
    ops.rotary_qkv_(qkv, cos, sin)
    # nn.MultiheadAttention: in_proj on q, k, v separately (they are different tensors), attention, out_proj
    qkv2 = qkv.view(M, 3, D)
    Wi, bi = P["attn.in_proj_weight"], P["attn.in_proj_bias"]
    pq = part_fwd(qkv2[:, 0], Wi, bi, 0, D).view(B, L, H, d)
    pk = part_fwd(qkv2[:, 1], Wi, bi, D, 2 * D).view(B, L, H, d)
    pv = part_fwd(qkv2[:, 2], Wi, bi, 2 * D, 3 * D).view(B, L, H, d)
    o = torch.empty(B, L, H, d, dtype=cdt, device=x.device)
    _, lse = ops.attn_fwd(_self_attn_views(pq, B, L, H, d, seq_axis), _self_attn_views(pk, B, L, H, d, seq_axis),
                          _self_attn_views(pv, B, L, H, d, seq_axis), out=_self_attn_views(o, B, L, H, d, seq_axis))
    ao = Lyr.linear_fwd(o.view(M, D), P["attn.out_proj.weight"], P["attn.out_proj.bias"])
    a = Lyr.linear_fwd(ao, P["attn_out.weight"], P["attn_out.bias"]).view(B, L, D)
    inv = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
    keep1 = Lyr.dropout_mask((B, L, D), p_drop, x.device) if p_drop > 0 else None
    x1 = ops.gate_residual_fwd(x, a, mod[:, 2 * D:3 * D], keep1, inv)
    st = dict(x=x, c=c, mod=mod, h1=h1, mean1=mean1, rstd1=rstd1, qkv=qkv, pq=pq, pk=pk, pv=pv, o=o, lse=lse, ao=ao,
              a=a, keep1=keep1, x1=x1, inv=inv, cos=cos, sin=sin, seq_axis=seq_axis, ca=None) if save else None
    xm = x1
    if other is not None and "cross_attn.norm.weight" in P:
        Lo = other.shape[1]
        Wc, bc = P["cross_attn.cross_attn.in_proj_weight"], P["cross_attn.cross_attn.in_proj_bias"]
        xn, meanc, rstdc = ops.ln_modulate_fwd(x1, P["cross_attn.norm.weight"].detach(),
                                               bias=P["cross_attn.norm.bias"].detach(), out_dtype=cdt)
        ctx = other.contiguous() if cdt == torch.float32 else ops.cast(other, cdt)
        cq = part_fwd(xn.view(M, D), Wc, bc, 0, D).view(B, L, H, d)
        ckv = part_fwd(ctx.view(B * Lo, D), Wc, bc, D, 3 * D).view(B, Lo, 2, H, d)
        km = None if other_mask is None else other_mask.to(torch.uint8).contiguous()
        co, clse = ops.attn_fwd(cq, ckv[:, :, 0], ckv[:, :, 1], key_mask=km)
        xc = Lyr.linear_fwd(co.view(M, D), P["cross_attn.cross_attn.out_proj.weight"],
                            P["cross_attn.cross_attn.out_proj.bias"], out_dtype=torch.float32).view(B, L, D)
        keepc = Lyr.dropout_mask((B, L, D), p_drop, x.device) if p_drop > 0 else None
        ones = _ones(B, D, x.device)
        xm = ops.gate_residual_fwd(x1, xc, ones, keepc, inv)  # x + dropout(x_ca): gate == 1
        if save:
            st["ca"] = dict(xn=xn, meanc=meanc, rstdc=rstdc, ctx=ctx, cq=cq, ckv=ckv, km=km, co=co, clse=clse, xc=xc,
                            keepc=keepc, ones=ones, Lo=Lo)
    h2, mean2, rstd2 = ops.ln_modulate_fwd(xm, P["norm2.weight"].detach(), shift=mod[:, 3 * D:4 * D],
                                           scale=mod[:, 4 * D:5 * D], out_dtype=cdt)
    if save:
        g, u = Lyr.linear_fwd(h2.view(M, D), P["mlp.0.weight"], P["mlp.0.bias"], gelu=True, want_pre=True)
    else:
        g, u = Lyr.linear_fwd(h2.view(M, D), P["mlp.0.weight"], P["mlp.0.bias"], gelu=True), None
    y = Lyr.linear_fwd(g, P["mlp.2.weight"], P["mlp.2.bias"]).view(B, L, D)
    keep2 = Lyr.dropout_mask((B, L, D), p_drop, x.device) if p_drop > 0 else None
    x2 = ops.gate_residual_fwd(xm, y, mod[:, 5 * D:6 * D], keep2, inv)
    if save:
        st.update(xm=xm, h2=h2, mean2=mean2, rstd2=rstd2, u=u, g=g, y=y, keep2=keep2)
    return x2, st


def cross_block_backward(dx2, st, P, n_heads):
    """-> (dx, dc, d_other or None, {param: grad or None})"""
    x = st["x"]
    B, L, D = x.shape
    M, H, d = B * L, n_heads, D // n_heads
    mod, cdt = st["mod"], st["h1"].dtype
    sink = GradSink()

    def G(name):
        return sink.get(name, P[name])

    dmod = torch.empty(B, 6 * D, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(dx2)  # = dx2 + the LayerNorm branches' gradients (dx2 is autograd's and stays untouched)
    # ---- MLP branch
    dy = ops.gate_residual_bwd(dx2, st["y"], mod[:, 5 * D:6 * D], dmod[:, 5 * D:6 * D], st["keep2"], st["inv"]).view(M, D)
    dg = part_bwd(st["g"], dy, P["mlp.2.weight"], P["mlp.2.bias"], 0, D, G("mlp.2.weight"), G("mlp.2.bias"))
    du = ops.gelu_bwd(st["u"], dg)
    dh2 = part_bwd(st["h2"].view(M, D), du, P["mlp.0.weight"], P["mlp.0.bias"], 0, 4 * D, G("mlp.0.weight"), G("mlp.0.bias"))
    ops.ln_modulate_bwd(dh2.view(B, L, D), st["xm"], P["norm2.weight"].detach(), mod[:, 4 * D:5 * D], st["mean2"],
                        st["rstd2"], dx, dx2, G("norm2.weight"), dmod[:, 3 * D:4 * D], dmod[:, 4 * D:5 * D])
    # ---- cross-attention branch: xm = x1 + dropout(out_proj(attn(q(LN(x1)), kv(other))))
    d_other = None
    ca = st["ca"]
    if ca is not None:
        Lo = ca["Lo"]
        Wc, bc = P["cross_attn.cross_attn.in_proj_weight"], P["cross_attn.cross_attn.in_proj_bias"]
        gWc, gbc = G("cross_attn.cross_attn.in_proj_weight"), G("cross_attn.cross_attn.in_proj_bias")
        dxc = ops.gate_residual_bwd(dx, ca["xc"], ca["ones"], None, ca["keepc"], st["inv"]).view(M, D)
        dxc_c = dxc if cdt == torch.float32 else ops.cast(dxc, cdt)
        dco = part_bwd(ca["co"].view(M, D), dxc_c, P["cross_attn.cross_attn.out_proj.weight"],
                       P["cross_attn.cross_attn.out_proj.bias"], 0, D, G("cross_attn.cross_attn.out_proj.weight"),
                       G("cross_attn.cross_attn.out_proj.bias"))
        dcq = torch.empty_like(ca["cq"])
        dckv = torch.empty_like(ca["ckv"])
        ops.attn_bwd(ca["cq"], ca["ckv"][:, :, 0], ca["ckv"][:, :, 1], ca["co"], dco.view(B, L, H, d), ca["clse"], dcq,
                     dckv[:, :, 0], dckv[:, :, 1], key_mask=ca["km"])
        dctx = part_bwd(ca["ctx"].view(B * Lo, D), dckv.view(B * Lo, 2 * D), Wc, bc, D, 3 * D, gWc, gbc,
                        dx_dtype=torch.float32)
        d_other = dctx.view(B, Lo, D)
        dxn = part_bwd(ca["xn"].view(M, D), dcq.view(M, D), Wc, bc, 0, D, gWc, gbc)
        dbias_b = torch.empty(B, D, dtype=torch.float32, device=x.device)
        ops.ln_modulate_bwd(dxn.view(B, L, D), st["x1"], P["cross_attn.norm.weight"].detach(), None, ca["meanc"],
                            ca["rstdc"], dx, True, G("cross_attn.norm.weight"), dbias_b, None)
        ops.colsum(dbias_b, out=G("cross_attn.norm.bias"), accumulate=True)
    # ---- self-attention branch (through nn.MultiheadAttention's projections)
    da = ops.gate_residual_bwd(dx, st["a"], mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], st["keep1"], st["inv"]).view(M, D)
    dao = part_bwd(st["ao"], da, P["attn_out.weight"], P["attn_out.bias"], 0, D, G("attn_out.weight"), G("attn_out.bias"))
    do = part_bwd(st["o"].view(M, D), dao, P["attn.out_proj.weight"], P["attn.out_proj.bias"], 0, D,
                  G("attn.out_proj.weight"), G("attn.out_proj.bias")).view(B, L, H, d)
    sa = st["seq_axis"]
    dpq, dpk, dpv = torch.empty_like(st["pq"]), torch.empty_like(st["pk"]), torch.empty_like(st["pv"])
    V_ = lambda t: _self_attn_views(t, B, L, H, d, sa)  # noqa: E731
    ops.attn_bwd(V_(st["pq"]), V_(st["pk"]), V_(st["pv"]), V_(st["o"]), V_(do), st["lse"], V_(dpq), V_(dpk), V_(dpv))
    Wi, bi = P["attn.in_proj_weight"], P["attn.in_proj_bias"]
    gWi, gbi = G("attn.in_proj_weight"), G("attn.in_proj_bias")
    qkv2 = st["qkv"].view(M, 3, D)
    dqkv = torch.empty_like(st["qkv"])
    dq2 = dqkv.view(M, 3, D)
    for slot, dp in enumerate((dpq, dpk, dpv)):
        r0 = slot * D
        part_bwd(qkv2[:, slot], dp.view(M, D), Wi, bi, r0, r0 + D, gWi, gbi, need_dx=False)
        part_dgrad(dp.view(M, D), Wi, r0, r0 + D, out=dq2[:, slot])
    ops.rotary_qkv_(dqkv, st["cos"], st["sin"], conj=True)
    dh1 = part_bwd(st["h1"].view(M, D), dqkv.view(M, 3 * D), P["attn_qkv.weight"], P["attn_qkv.bias"], 0, 3 * D,
                   G("attn_qkv.weight"), G("attn_qkv.bias"))
    ops.ln_modulate_bwd(dh1.view(B, L, D), x, P["norm1.weight"].detach(), mod[:, D:2 * D], st["mean1"], st["rstd1"], dx,
                        True, G("norm1.weight"), dmod[:, :D], dmod[:, D:2 * D])
    # ---- adaLN
    # (fp32 operands whatever the compute dtype, as in layers.ddit_block_backward: [B, 6D] and [B, C] are fp32 as they stand)
    dc = part_bwd(st["c"], dmod, P["adaLN_modulation.weight"], P["adaLN_modulation.bias"], 0, 6 * D,
                  G("adaLN_modulation.weight"), G("adaLN_modulation.bias"))
    return dx, dc, d_other, sink.result(list(P.keys()))
