"""Drop-in for protlig_dd/model/rotary.py: the cos/sin cache (built once per sequence length exactly as the
reference builds it, rotary.py:13-27) and apply_rotary_pos_emb on the HIP rotary kernel."""
import torch

from .. import ops


class Rotary(torch.nn.Module):
    def __init__(self, dim, base=10_000):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)
        self.seq_len_cached = None
        self.cos_cached = None
        self.sin_cached = None
        self._by_len = {}  # (L, device) -> (cos, sin): both tracks of the dual model keep their tables

    def forward(self, x, seq_dim=1):
        seq_len = x.shape[seq_dim]
        hit = self._by_len.get((seq_len, x.device))
        if hit is not None:
            self.seq_len_cached, self.cos_cached, self.sin_cached = seq_len, hit[0], hit[1]
            return hit
        if seq_len != self.seq_len_cached or self.cos_cached.device != x.device:
            self.seq_len_cached = seq_len
            t = torch.arange(seq_len, device=x.device).type_as(self.inv_freq)
            freqs = torch.einsum("i,j->ij", t, self.inv_freq.clone().to(x.device))
            emb = torch.cat((freqs, freqs), dim=-1)
            # dims are: batch, seq_len, qkv, head, dim; the v slot is the identity rotation
            self.cos_cached = emb.cos()[None, :, None, None, :].repeat(1, 1, 3, 1, 1)
            self.sin_cached = emb.sin()[None, :, None, None, :].repeat(1, 1, 3, 1, 1)
            self.cos_cached[:, :, 2, :, :].fill_(1.)
            self.sin_cached[:, :, 2, :, :].fill_(0.)
            half = emb.shape[-1] // 2
            # compact [L, d/2] fp32 halves the kernel reads (what flash-attn's rotary takes, rotary.py:44-45)
            self.cos_cached._sedd_compact = (emb.cos()[:, :half].float().contiguous(),
                                             emb.sin()[:, :half].float().contiguous())
            self._by_len[(seq_len, x.device)] = (self.cos_cached, self.sin_cached)
        return self.cos_cached, self.sin_cached


def apply_rotary_pos_emb(qkv, cos, sin):
    """rotary.py:42-51 (out of place like the TorchScript fallback): qkv [B, L, 3, H, d]."""
    comp = getattr(cos, "_sedd_compact", None)
    if comp is None:
        d = cos.shape[-1]
        comp = (cos[0, :, 0, 0, :d // 2].float().contiguous(), sin[0, :, 0, 0, :d // 2].float().contiguous())
    return ops.rotary_qkv_(qkv.contiguous().clone(), comp[0], comp[1])
