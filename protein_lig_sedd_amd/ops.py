"""Thin tensor-level wrappers over the C ABI (include/sedd_hip.h). Each takes/returns torch CUDA tensors, launches on
torch's current stream and allocates outputs with torch's caching allocator. No arithmetic happens in Python."""
import torch

from . import _lib as _L
from ._lib import ABSORB, BF16, F32, UNIFORM, check, lib  # noqa: F401


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype} (fp32 and bf16 only)")


def _tdt(code):
    return torch.float32 if code == F32 else torch.bfloat16


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("protein_lig_sedd_amd ops need CUDA(HIP) tensors: there is no CPU fallback")


def _c(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _batch_scalar(s, B):
    """sigma-like argument -> contiguous fp32 [B] (accepts [B], [B,1], scalar)."""
    s = s.reshape(-1).to(torch.float32)
    if s.numel() == 1 and B != 1:
        s = s.expand(B)
    if s.numel() != B:
        raise ValueError(f"expected {B} per-batch scalars, got {s.numel()}")
    return s.contiguous()


# ------------------------------------------------------------------------------------------------ graph ops
def sample_transition(graph, x0, sigma, u, r=None):
    _cuda(x0, sigma, u, r)
    x0 = _c(x0, torch.int64)
    B = x0.shape[0] if x0.dim() > 1 else 1
    M = x0.numel()
    L = M // max(B, 1) if M else 1
    out = torch.empty_like(x0)
    check(lib().sedd_sample_transition(graph.code, _p(x0), _p(_batch_scalar(sigma, B)), _p(_c(u, torch.float32)),
                                       _p(_c(r, torch.int64)), _p(out), M, max(L, 1), graph.dim, _stream()),
          "sedd_sample_transition")
    return out


def _rows(x):
    B = x.shape[0] if x.dim() > 1 else 1
    M = x.numel()
    return B, M, (M // B if M else 1)


def score_entropy_fwd(graph, score, sigma, x, x0, gw=None):
    """-> (out fp32 like x, dscore or None). gw: per-batch gradient weights [B] for the fused loss path."""
    _cuda(score, sigma, x, x0)
    score, x, x0 = _c(score), _c(x, torch.int64), _c(x0, torch.int64)
    B, M, L = _rows(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    ds = torch.empty_like(score) if gw is not None else None
    check(lib().sedd_score_entropy_fwd(graph.code, _p(score), _dt(score), _p(_batch_scalar(sigma, B)), _p(x), _p(x0),
                                       _p(out), _p(ds), _p(None if gw is None else _batch_scalar(gw, B)), M, L,
                                       score.shape[-1], _stream()), "sedd_score_entropy_fwd")
    return out, ds


def score_entropy_bwd(graph, score, sigma, x, x0, gout):
    score, x, x0 = _c(score), _c(x, torch.int64), _c(x0, torch.int64)
    B, M, L = _rows(x)
    ds = torch.empty_like(score)
    check(lib().sedd_score_entropy_bwd(graph.code, _p(score), _dt(score), _p(_batch_scalar(sigma, B)), _p(x), _p(x0),
                                       _p(_c(gout, torch.float32)), _p(ds), M, L, score.shape[-1], _stream()),
          "sedd_score_entropy_bwd")
    return ds


def gumbel_argmax(p, u):
    _cuda(p, u)
    p, u = _c(p), _c(u, torch.float32)
    if p.shape != u.shape:
        raise ValueError(f"probs {tuple(p.shape)} and noise {tuple(u.shape)} differ")
    V = p.shape[-1]
    out = torch.empty(p.shape[:-1], dtype=torch.int64, device=p.device)
    check(lib().sedd_gumbel_argmax(_p(p), _dt(p), _p(u), _p(out), out.numel(), V, _stream()), "sedd_gumbel_argmax")
    return out


def predictor_step(graph, step, logscore, x, coef, u):
    _cuda(logscore, x, coef, u)
    logscore, x, u = _c(logscore), _c(x, torch.int64), _c(u, torch.float32)
    B, M, L = _rows(x)
    V = logscore.shape[-1]
    Vu = V - 1 if (graph.code == ABSORB and step == _L.STEP_DENOISE) else V
    if u.shape[-1] != Vu or u.numel() != M * Vu:
        raise ValueError(f"noise tensor must be [..., {Vu}] with {M} rows, got {tuple(u.shape)}")
    out = torch.empty_like(x)
    check(lib().sedd_predictor_step(graph.code, step, _p(logscore), _dt(logscore), _p(x), _p(_batch_scalar(coef, B)),
                                    _p(u), _p(out), M, L, V, _stream()), "sedd_predictor_step")
    return out


def transp_rate(graph, x):
    x = _c(x, torch.int64)
    out = torch.empty(*x.shape, graph.dim, dtype=torch.float32, device=x.device)
    check(lib().sedd_transp_rate(graph.code, _p(x), _p(out), x.numel(), graph.dim, _stream()), "sedd_transp_rate")
    return out


def reverse_rate(graph, x, score):
    x, score = _c(x, torch.int64), _c(score, torch.float32)
    out = torch.empty_like(score)
    check(lib().sedd_reverse_rate(graph.code, _p(x), _p(score), _p(out), x.numel(), score.shape[-1], _stream()),
          "sedd_reverse_rate")
    return out


def staggered_score(graph, score, dsigma):
    score = _c(score, torch.float32)
    B = score.shape[0]
    M = score.numel() // score.shape[-1]
    out = torch.empty_like(score)
    check(lib().sedd_staggered_score(graph.code, _p(score), _p(_batch_scalar(dsigma, B)), _p(out), M, M // B,
                                     score.shape[-1], _stream()), "sedd_staggered_score")
    return out


def transp_transition(graph, x, sigma):
    x = _c(x, torch.int64)
    B, M, L = _rows(x)
    out = torch.empty(*x.shape, graph.dim, dtype=torch.float32, device=x.device)
    check(lib().sedd_transp_transition(graph.code, _p(x), _p(_batch_scalar(sigma, B)), _p(out), M, L, graph.dim,
                                       _stream()), "sedd_transp_transition")
    return out
