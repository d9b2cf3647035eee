"""Thin tensor-level wrappers over the C ABI (include/sedd_hip.h). Each takes/returns torch CUDA tensors, launches on
torch's current stream and allocates outputs with torch's caching allocator. No arithmetic happens in Python."""
import collections
import os
import threading

import torch

from . import _lib as _L
from ._lib import ABSORB, BF16, F32, UNIFORM, check, lib  # noqa: F401


class KernelProf:
    """Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline figures).
    stride > 1 times every stride-th launch of each family only: an event pair costs ~3 us of stream time (the
    launches on either side can no longer overlap), 1.2 ms per cfg-2 step when every launch is bracketed."""

    def __init__(self, stride=1):
        self.stride = max(int(stride), 1)
        self.recs, self.seen = {}, {}

    def want(self, name):
        n = self.seen.get(name, 0)
        self.seen[name] = n + 1
        # the phase moves on by one every 97 launches, so a family whose launches per step are a multiple of the stride
        # is still sampled at every position of the step over a run
        return (n + n // 97) % self.stride == 0

    def add(self, name, start, end, work):
        self.recs.setdefault(name, []).append((start, end, work))

    def summary(self):
        """name -> (launches seen, estimated total ms, total work): the timed sample scaled to all launches; call
        after a device synchronize."""
        out = {}
        for name, rs in self.recs.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in rs)
            work = sum(w for _, _, w in rs)
            k = self.seen[name] / len(rs)
            out[name] = (self.seen[name], ms * k, work * k)
        return out


PROF = None  # set to a KernelProf to time instrumented launches


class _Timed:
    def __init__(self, name, work):
        self.name, self.work = name, work
        self.on = False

    def __enter__(self):
        if PROF is not None and PROF.want(self.name):
            self.on = True
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            PROF.add(self.name, self.s, self.e, self.work)
        return False


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


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """raw handle of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object
    through several Python layers — ~8 us a call, a fifth of the host time of a launch-bound step.)"""
    ov = getattr(_TLS, "stream", None)
    if ov is not None:
        return ov
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


_TLS = threading.local()  # .stream: raw handle every launch of THIS THREAD goes to while a SideStream fork is open


class _RawFork:
    """`with` body: this module's launches go to the side stream.  torch's own current stream is left alone — entering
    torch.cuda.stream() costs tens of microseconds, which 290 forks a step turned into the step's bottleneck — so the body
    must launch through this module only and allocate nothing that is freed before join()."""
    __slots__ = ("raw",)

    def __init__(self, raw):
        self.raw = raw

    def __enter__(self):
        _TLS.stream = self.raw

    def __exit__(self, *exc):
        _TLS.stream = None
        return False


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("protein_lig_sedd_amd ops need CUDA(HIP) tensors: there is no CPU fallback")


class SideStream:
    """A second launch stream for work that is OFF the critical path of backward — weight gradients and bias-gradient column
    sums are consumed by the optimiser only, while the data gradients feed the next layer's backward.  When the kernels of a
    step each fill a fraction of the chip (the dual-track model at batch 4: 64-128 tiles per GEMM on 256 CUs), the two chains
    run side by side instead of one after the other.  `fork(keep...)`: the side stream waits for everything launched so far
    on the current stream; the tensors it will read are kept alive until the side stream is done with them (the caching
    allocator would otherwise hand their memory to a later allocation of the main stream while the side stream still reads
    them): each flushed batch is followed by an event on the side stream and released once that event has completed — so
    what is held is a few batches, not a whole backward pass of activations and gradients — or at `join()` at the latest.
    `join()`: the current stream waits for the side stream.  Capture-safe (events inside a capture become graph edges; no
    event is queried while capturing, the tensors then stay until join)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.keep = []
        self.dirty = False
        self._events = [torch.cuda.Event() for _ in range(32)]  # re-recorded in turn: creating one per fork is slow
        self._next = 0
        self._raw = _RawFork(self.stream.cuda_stream)
        self.pending = []
        self.batch = int(os.environ.get("SEDD_SIDE_BATCH", "16"))
        self._inflight = collections.deque()  # (event on the side stream behind a flushed batch, the tensors it reads)
        self._done_pool = []

    def fork(self, *tensors, torch_ops=False):
        """torch_ops=True: the body also runs torch kernels or allocates temporaries (the slow torch.cuda.stream context);
        otherwise only this module's launches move over."""
        ev = self._events[self._next]
        self._next = (self._next + 1) % len(self._events)
        ev.record()
        self.stream.wait_event(ev)
        self.keep.extend(t for t in tensors if t is not None)
        self.dirty = True
        return torch.cuda.stream(self.stream) if torch_ops else self._raw

    def defer(self, fn, *tensors, torch_ops=False):
        """queue `fn` for the side stream; the queue is flushed — ONE fork for all of it — every `batch` entries and at
        join().  An event record + stream wait per weight gradient (290 a step in the dual-track model) cost more host
        time than the launches themselves; the deferred launches lose nothing, they are off the critical path."""
        self.pending.append((fn, torch_ops, tensors))
        if len(self.pending) >= self.batch:
            self.flush()

    def flush(self):
        if not self.pending:
            return
        todo, self.pending = self.pending, []
        with self.fork():
            for fn, torch_ops, _ in todo:
                if torch_ops:
                    _TLS.stream = None
                    with torch.cuda.stream(self.stream):
                        fn()
                    _TLS.stream = self._raw.raw
                else:
                    fn()
        held = [t for _, _, ts in todo for t in ts if t is not None]
        if torch.cuda.is_current_stream_capturing():
            self.keep.extend(held)
            return
        ev = self._done_pool.pop() if self._done_pool else torch.cuda.Event()
        ev.record(self.stream)
        self._inflight.append((ev, held))
        while len(self._inflight) > 1 and self._inflight[0][0].query():
            self._done_pool.append(self._inflight.popleft()[0])

    def join(self):
        self.flush()
        if self.dirty:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
            self.dirty = False
        self.keep = []
        while self._inflight:
            self._done_pool.append(self._inflight.popleft()[0])


SIDE = None  # set to a SideStream by a trainer for the duration of a backward pass

_SCRATCH = {}
_SCRATCH_RETIRED = []


def _scratch(kind, device, need):
    """fp32 scratch of >= `need` floats for the cross-block reductions (fixed-order folds instead of atomics), one per
    (kind, device, STREAM): launches on one stream are ordered, so consecutive calls share it; two streams on one GPU (a
    training step beside an evaluation forward) get buffers of their own instead of racing on one.  A buffer that has to
    grow is RETIRED, not freed — a HIP graph captured earlier may still hold its address."""
    key = (kind, device.index, _stream())
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _SCRATCH_RETIRED.append(ws)
        ws = _SCRATCH[key] = torch.empty(need, dtype=torch.float32, device=device)
    return ws


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
    V = score.shape[-1]
    with _Timed("score_entropy", float(M) * ((2 if gw is not None else 1) * V * score.element_size() + 20)):
        check(lib().sedd_score_entropy_fwd(graph.code, _p(score), _dt(score), _p(_batch_scalar(sigma, B)), _p(x),
                                           _p(x0), _p(out), _p(ds), _p(None if gw is None else _batch_scalar(gw, B)),
                                           M, L, V, _stream()), "sedd_score_entropy_fwd")
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
    with _Timed("predictor_step", float(M) * (V * logscore.element_size() + Vu * 4 + 16)):
        check(lib().sedd_predictor_step(graph.code, step, _p(logscore), _dt(logscore), _p(x),
                                        _p(_batch_scalar(coef, B)), _p(u), _p(out), M, L, V, _stream()),
              "sedd_predictor_step")
    return out


class PhiloxNoise:
    """Stands in for a predictor update's uniform tensor `u`: the kernel generates it (Philox4x32-10 keyed by `seed`,
    counter (row, vocabulary slot / 4, offset) — include/sedd_hip.h, sedd_predictor_step_philox) instead of reading a
    torch.rand tensor.  The reference draws that tensor with torch.rand_like in utils/catsample.py:12."""
    __slots__ = ("seed", "offset")

    def __init__(self, seed, offset=0):
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset)
        if not 0 <= self.offset < 2 ** 32:
            raise ValueError("PhiloxNoise offset must fit 32 bits")


class PhiloxStream:
    """`noise_seq` for get_pc_sampler: update i draws its noise in-kernel with PhiloxNoise(seed, i)."""

    def __init__(self, seed):
        self.seed = int(seed)

    def __getitem__(self, i):
        return PhiloxNoise(self.seed, i)


def predictor_step_philox(graph, step, logscore, x, coef, noise):
    _cuda(logscore, x, coef)
    logscore, x = _c(logscore), _c(x, torch.int64)
    B, M, L = _rows(x)
    V = logscore.shape[-1]
    out = torch.empty_like(x)
    seed = noise.seed - (1 << 64) if noise.seed >= (1 << 63) else noise.seed  # (the binding passes int64)
    with _Timed("predictor_step_philox", float(M) * (V * logscore.element_size() + 16)):
        rc = lib().sedd_predictor_step_philox(graph.code, step, _p(logscore), _dt(logscore), _p(x),
                                              _p(_batch_scalar(coef, B)), seed, noise.offset, _p(out), M, L, V, _stream())
    if rc == -2:
        # shapes the row-per-lane kernel does not take (the ligand vocabulary, unaligned views): the same uniforms as a
        # tensor, then the explicit-noise kernel — identical results, one more pass over [M, Vu]
        Vu = V - 1 if (graph.code == ABSORB and step == _L.STEP_DENOISE) else V
        u = torch.empty(*x.shape, Vu, dtype=torch.float32, device=x.device)
        check(lib().sedd_philox_uniform(seed, noise.offset, _p(u), M, Vu, _stream()), "sedd_philox_uniform")
        return predictor_step(graph, step, logscore, x, coef, u)
    check(rc, "sedd_predictor_step_philox")
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


# ------------------------------------------------------------------------------------------- model: pointwise
def timestep_embedding(sigma, dim=256):
    sigma = _c(sigma.reshape(-1), torch.float32)
    out = torch.empty(sigma.numel(), dim, dtype=torch.float32, device=sigma.device)
    check(lib().sedd_timestep_embedding(_p(sigma), _p(out), sigma.numel(), dim, _stream()), "sedd_timestep_embedding")
    return out


def cond_combine_fwd(c, terms, training, momentum=0.1, eps=1e-5):
    """SEDD's pooled conditioners (transformer.py:289-301): mean([c, BN(x_e), BN(x_m)]).  terms: two entries (ESM,
    MolFormer), each None or (x [B, C], weight, bias, running_mean, running_var); the running buffers are updated in
    place in training mode.  -> (out [B, C], saved [(mean, rstd) or None] x 2)."""
    _cuda(c)
    c = _c(c, torch.float32)
    B, C = c.shape
    if training and B < 2 and any(t is not None for t in terms):  # torch's nn.BatchNorm1d error
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {torch.Size([B, C])}")
    out = torch.empty_like(c)
    args, saved, keep = [], [], []
    for t in terms:
        if t is None:
            args += [None] * 7
            saved.append(None)
            continue
        x, w, b, rm, rv = t
        x = _c(x, torch.float32)
        keep.append(x)
        st = torch.empty(2, C, dtype=torch.float32, device=c.device)
        saved.append((x, st[0], st[1]))
        args += [_p(x), _p(w), _p(b), _p(rm), _p(rv), _p(st[0]), _p(st[1])]
    check(lib().sedd_cond_combine_fwd(_p(c), *args, _p(out), 1 if training else 0, float(momentum), float(eps), B, C,
                                      _stream()), "sedd_cond_combine_fwd")
    return out, saved


def cond_combine_bwd(dout, saved, weights, sinks, training, need_dx=(True, True)):
    """-> (dc, [dx or None] x 2, [(dw, db) or None] x 2).  saved: from cond_combine_fwd; weights: BatchNorm weights;
    sinks: per term (dw_buffer, db_buffer) to ACCUMULATE into, or None for fresh tensors (all terms alike)."""
    dout = _c(dout, torch.float32)
    B, C = dout.shape
    dc = torch.empty_like(dout)
    args, dxs, dws = [], [], []
    accumulate = any(s is not None for s in sinks)
    for sv, w, sink, ndx in zip(saved, weights, sinks, need_dx):
        if sv is None:
            args += [None] * 7
            dxs.append(None)
            dws.append(None)
            continue
        x, mean, rstd = sv
        dx = torch.empty_like(x) if ndx else None
        if sink is None:
            z = torch.zeros(2, C, dtype=torch.float32, device=dout.device) if accumulate else \
                torch.empty(2, C, dtype=torch.float32, device=dout.device)
            dw, db = z[0], z[1]
            dws.append((dw, db))
        else:
            dw, db = sink
            dws.append(None)
        dxs.append(dx)
        args += [_p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw), _p(db)]
    check(lib().sedd_cond_combine_bwd(_p(dout), *args, _p(dc), 1 if training else 0, 1 if accumulate else 0, B, C,
                                      _stream()), "sedd_cond_combine_bwd")
    return dc, dxs, dws


def silu_fwd(x):
    x = _c(x, torch.float32)
    y = torch.empty_like(x)
    check(lib().sedd_silu_fwd(_p(x), _p(y), x.numel(), _stream()), "sedd_silu_fwd")
    return y


def silu_bwd(x, dy):
    x, dy = _c(x, torch.float32), _c(dy, torch.float32)
    dx = torch.empty_like(x)
    check(lib().sedd_silu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "sedd_silu_bwd")
    return dx


def embedding_fwd(table, idx):
    idx = _c(idx, torch.int64)
    V, D = table.shape
    out = torch.empty(*idx.shape, D, dtype=torch.float32, device=table.device)
    check(lib().sedd_embedding_fwd(_p(table), _p(idx), _p(out), idx.numel(), D, V, _stream()), "sedd_embedding_fwd")
    return out


def embedding_bwd(dout, idx, dtable):
    """dtable[idx] += dout (in place on dtable, fp32 [V, D])."""
    idx, dout = _c(idx, torch.int64), _c(dout, torch.float32)
    V, D = dtable.shape
    need = lib().sedd_embedding_bwd_ws_floats(idx.numel(), D, V)  # fixed-order fold instead of atomics where supported
    ws = None
    if need and not _ATOMIC:
        ws = _scratch("emb", dtable.device, need)
    check(lib().sedd_embedding_bwd(_p(dout), _p(idx), _p(dtable), _p(ws), idx.numel(), D, V, _stream()),
          "sedd_embedding_bwd")


def _mod_args(t):
    """(pointer-able tensor, dtype code, row stride) of a [B, D] view into the adaLN output."""
    if t is None:
        return None, F32, 0
    if t.stride(-1) != 1:
        raise ValueError("modulation slice must be contiguous in its last dim")
    return t, _dt(t), t.stride(0)


def ln_modulate_fwd(x, w, shift=None, scale=None, bias=None, out_dtype=torch.float32):
    """x fp32 [B, L, D] -> (y [B, L, D] out_dtype, mean [M], rstd [M]); shift/scale are [B, D] (strided) views."""
    B, L, D = x.shape
    M = B * L
    y = torch.empty(B, L, D, dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    sc, mdt, ms = _mod_args(scale)
    sh, _, _ = _mod_args(shift)
    check(lib().sedd_ln_modulate_fwd(_p(x), _p(w), _p(bias), _p(sh), _p(sc), mdt, ms, _p(y), _dt(y), _p(mean),
                                     _p(rstd), M, L, D, _stream()), "sedd_ln_modulate_fwd")
    return y, mean, rstd


def ln_modulate_bwd(dy, x, w, scale, mean, rstd, dx, accumulate, dw, dshift, dscale):
    """dx fp32 [B,L,D] = LN-modulate gradient (+ accumulate): accumulate False -> written, True -> dx += (in place),
    a tensor -> dx = accumulate + gradient (no copy of the incoming residual gradient needed);
    dw fp32 [D] accumulated (+=); dshift/dscale fp32 [B, D] (strided) views written (may be None)."""
    add = dx if accumulate is True else (None if (accumulate is False or accumulate is None) else accumulate)
    if add is not None and (add.shape != dx.shape or add.dtype != torch.float32 or not add.is_contiguous()):
        raise ValueError("ln_modulate_bwd: addend must be a contiguous fp32 tensor of dx's shape")
    B, L, D = x.shape
    M = B * L
    nchunk = lib().sedd_ln_bwd_chunks(M, L)
    ws = torch.empty(3 * nchunk * D, dtype=torch.float32, device=x.device)
    sc, mdt, ms = _mod_args(scale)
    check(lib().sedd_ln_modulate_bwd(_p(dy), _dt(dy), _p(x), _p(w), _p(sc), mdt, ms, _p(mean), _p(rstd), _p(dx),
                                     _p(add), _p(ws), M, L, D, _stream()), "sedd_ln_modulate_bwd")
    dstride = dshift.stride(0) if dshift is not None else (dscale.stride(0) if dscale is not None else D)
    check(lib().sedd_ln_modulate_bwd_reduce(_p(ws), _p(dw), _p(dshift), _p(dscale), dstride, M, L, D, _stream()),
          "sedd_ln_modulate_bwd_reduce")


def ln_modulate_bwd_gate(dy, x, w, scale, mean, rstd, dx, accumulate, dw, dshift, dscale, a, gate, dgate, keep, inv_keep,
                         dbias=None):
    """ln_modulate_bwd + gate_residual_bwd(dx, a, gate, dgate, keep, inv_keep[, dbias]) in one pass over the rows -> da.
    (x = residual + gate * dropout(a) is the gated residual that feeds this LayerNorm; keep: DropSeed or None;
    dbias fp32 [D] (optional) += da.sum over rows, as gate_residual_bwd's.)"""
    add = dx if accumulate is True else (None if (accumulate is False or accumulate is None) else accumulate)
    if add is not None and (add.shape != dx.shape or add.dtype != torch.float32 or not add.is_contiguous()):
        raise ValueError("ln_modulate_bwd_gate: addend must be a contiguous fp32 tensor of dx's shape")
    if a.dtype != dy.dtype or a.shape != dy.shape or not a.is_contiguous():
        raise ValueError("ln_modulate_bwd_gate: a must match dy in dtype and shape")
    B, L, D = x.shape
    M = B * L
    nchunk = lib().sedd_ln_bwd_chunks(M, L)
    ws = torch.empty((5 if dbias is not None else 4) * nchunk * D, dtype=torch.float32, device=x.device)
    wsg = ws[3 * nchunk * D:]
    sc, mdt, ms = _mod_args(scale)
    g, gdt, gs = _mod_args(gate)
    if (gdt, gs) != (mdt, ms):
        raise ValueError("scale and gate must be slices of modulation tensors of one dtype and row stride")
    _, seed, pd = _keep_args(keep)
    da = torch.empty_like(a)
    fn = lib().sedd_ln_modulate_bwd_gate if dbias is None else lib().sedd_ln_modulate_bwd_gate_colsum
    check(fn(_p(dy), _dt(dy), _p(x), _p(w), _p(sc), mdt, ms, _p(mean), _p(rstd), _p(dx), _p(add), _p(ws), _p(a), _p(g),
             float(inv_keep), seed, pd, _p(da), _p(wsg), M, L, D, _stream()), "sedd_ln_modulate_bwd_gate")
    dstride = dshift.stride(0) if dshift is not None else (dscale.stride(0) if dscale is not None else D)
    check(lib().sedd_ln_modulate_bwd_reduce(_p(ws), _p(dw), _p(dshift), _p(dscale), dstride, M, L, D, _stream()),
          "sedd_ln_modulate_bwd_reduce")
    check(lib().sedd_gate_residual_bwd_reduce(_p(wsg), _p(dgate), dgate.stride(0), _p(dbias), M, L, D, _stream()),
          "sedd_gate_residual_bwd_reduce")
    return da


class DropSeed:
    """Dropout by counter-based hash: (seed, p) stand in for a [M, D] keep-mask; forward and backward regenerate the
    same mask in the kernel (sedd_gate_residual_fwd / _bwd)."""
    __slots__ = ("seed", "p")
    _counter = 0
    # while a train step is being CAPTURED as a HIP graph: (device uint64 salt tensor, sites handed out so far) — the seeds
    # are then indirect (bit 63 | site << 48 | address of the salt, see sedd_gate_residual_fwd): the launch arguments the
    # capture freezes stay valid, and the masks change with the salt the trainer writes before every replay
    _indirect = None

    def __init__(self, p, seed=None):
        if seed is None and DropSeed._indirect is not None:
            salt, site = DropSeed._indirect
            site += 1
            if site >= 1 << 15:
                raise RuntimeError("more than 32767 dropout sites in one captured step")
            DropSeed._indirect = (salt, site)
            seed = (1 << 63) | (site << 48) | salt.data_ptr()
        if seed is None:
            DropSeed._counter += 1
            # the rank is mixed in: data-parallel ranks are seeded identically (same initial weights) but must draw
            # their own dropout masks, as the reference's DDP ranks do from their own device generators
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
            seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + DropSeed._counter * 0xD1B54A32D192ED03
                    + rank * 0xA24BAED4963EE407) & 0x7FFFFFFFFFFFFFFF
        self.seed, self.p = int(seed), float(p)


def _keep_args(keep):
    """-> (mask pointer, seed, p_drop) for a uint8 mask tensor, a DropSeed or None."""
    if isinstance(keep, DropSeed):
        return None, (keep.seed - (1 << 64) if keep.seed >> 63 else keep.seed), keep.p  # (ctypes int64 carries the bits)
    return _p(keep), 0, 0.0


def gate_residual_fwd(residual, a, gate, keep=None, inv_keep=1.0):
    B, L, D = residual.shape
    out = torch.empty_like(residual)
    g, mdt, ms = _mod_args(gate)
    kp, seed, pd = _keep_args(keep)
    check(lib().sedd_gate_residual_fwd(_p(residual), _p(a), _dt(a), _p(g), mdt, ms, kp, float(inv_keep), seed, pd,
                                       _p(out), B * L, L, D, _stream()), "sedd_gate_residual_fwd")
    return out


def gate_residual_ln_modulate_supported(residual, a, keep):
    return (residual.shape[-1] <= 1024 and (keep is None or isinstance(keep, DropSeed))
            and not os.environ.get("SEDD_NO_FUSED_GATE_LN"))


def gate_residual_ln_modulate_fwd(residual, a, gate, keep, inv_keep, w, shift, scale):
    """x1 = residual + gate * dropout(a); h = LN(x1) * w * (1 + scale) + shift in one pass over the rows.
    gate / shift / scale: [B, D] slices of one modulation tensor; keep: DropSeed or None.
    -> (x1 fp32, h in a's dtype, mean [M], rstd [M])."""
    B, L, D = residual.shape
    M = B * L
    g, mdt, ms = _mod_args(gate)
    sh, mdt2, ms2 = _mod_args(shift)
    sc, mdt3, ms3 = _mod_args(scale)
    if not (mdt == mdt2 == mdt3 and ms == ms2 == ms3):
        raise ValueError("gate / shift / scale must be slices of one modulation tensor")
    _, seed, pd = _keep_args(keep)
    x1 = torch.empty_like(residual)
    h = torch.empty(B, L, D, dtype=a.dtype, device=a.device)
    mean = torch.empty(M, dtype=torch.float32, device=a.device)
    rstd = torch.empty(M, dtype=torch.float32, device=a.device)
    check(lib().sedd_gate_residual_ln_modulate_fwd(_p(residual), _p(a), _dt(a), _p(g), _p(sh), _p(sc), mdt, ms,
                                                   float(inv_keep), seed, pd, _p(w), _p(x1), _p(h), _p(mean), _p(rstd),
                                                   M, L, D, _stream()), "sedd_gate_residual_ln_modulate_fwd")
    return x1, h, mean, rstd


def gate_residual_bwd(dout, a, gate, dgate, keep=None, inv_keep=1.0, dbias=None):
    """-> da (dtype of a); dgate fp32 [B, D] (strided view) written; dbias fp32 [D] (optional) += da.sum over rows
    (the bias gradient of the Linear that produced `a`, folded into the same pass)."""
    B, L, D = dout.shape
    M = B * L
    da = torch.empty_like(a)
    nchunk = lib().sedd_ln_bwd_chunks(M, L)
    ws = torch.empty((2 if dbias is not None else 1) * nchunk * D, dtype=torch.float32, device=dout.device)
    g, mdt, ms = _mod_args(gate)
    kp, seed, pd = _keep_args(keep)
    check(lib().sedd_gate_residual_bwd(_p(dout), _p(a), _dt(a), _p(g), mdt, ms, kp, float(inv_keep), seed, pd, _p(da),
                                       _p(ws), 0 if dbias is None else 1, M, L, D, _stream()), "sedd_gate_residual_bwd")
    if dgate is not None:  # (None: the gate is a constant — the cross-attention residual's all-ones gate — and has no gradient)
        check(lib().sedd_gate_residual_bwd_reduce(_p(ws), _p(dgate), dgate.stride(0), _p(dbias), M, L, D, _stream()),
              "sedd_gate_residual_bwd_reduce")
    elif dbias is not None:
        raise ValueError("gate_residual_bwd: dbias rides on the gate gradient's fold")
    return da


def rotary_qkv_(qkv, cos, sin, conj=False):
    """in place on qkv [B, L, 3, H, d]; cos/sin fp32 [L, d/2]."""
    B, L, three, H, d = qkv.shape
    assert three == 3 and qkv.is_contiguous()
    check(lib().sedd_rotary_qkv(_p(qkv), _dt(qkv), _p(cos), _p(sin), B, L, H, d, 1 if conj else 0, _stream()),
          "sedd_rotary_qkv")
    return qkv


def gelu_fwd(u):
    g = torch.empty_like(u)
    check(lib().sedd_gelu_fwd(_p(u), _p(g), _dt(u), u.numel(), _stream()), "sedd_gelu_fwd")
    return g


def gelu_bwd(u, dg):
    du = torch.empty_like(u)
    check(lib().sedd_gelu_bwd(_p(u), _p(dg), _p(du), _dt(u), u.numel(), _stream()), "sedd_gelu_bwd")
    return du


def logits_epilogue_(logits, sigma, idx, scale_by_sigma=True):
    B = logits.shape[0]
    V = logits.shape[-1]
    M = logits.numel() // V
    check(lib().sedd_logits_epilogue(_p(logits), _dt(logits), _p(_batch_scalar(sigma, B)), _p(_c(idx, torch.int64)),
                                     1 if scale_by_sigma else 0, M, M // B, V, _stream()), "sedd_logits_epilogue")
    return logits


def logits_epilogue_bwd_(dlogits, idx):
    V = dlogits.shape[-1]
    check(lib().sedd_logits_epilogue_bwd(_p(dlogits), _dt(dlogits), _p(_c(idx, torch.int64)), dlogits.numel() // V, V,
                                         _stream()), "sedd_logits_epilogue_bwd")
    return dlogits


def colsum(a, out=None, accumulate=False):
    """a [M, N] (row stride allowed) -> fp32 [N]; accumulate into `out` when given and accumulate=True."""
    M, N = a.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
        accumulate = False
    ws = None
    if M > 128 and not _ATOMIC:  # several row blocks: fixed-order fold out of a scratch instead of atomics
        ws = _scratch("colsum_rows", a.device, ((M + 127) // 128) * N)
    check(lib().sedd_colsum(_p(a), _dt(a), a.stride(0), _p(out), _p(ws), 1 if accumulate else 0, M, N, _stream()),
          "sedd_colsum")
    return out


def transpose(a, out_dtype=None):
    """a [M, N] (row stride allowed) -> contiguous [N, M] in out_dtype."""
    M, N = a.shape
    out = torch.empty(N, M, dtype=out_dtype or a.dtype, device=a.device)
    check(lib().sedd_transpose(_p(a), _dt(a), a.stride(0), _p(out), _dt(out), M, M, N, _stream()), "sedd_transpose")
    return out


def cast(a, dtype, out=None):
    a = _c(a)
    if out is None:
        out = torch.empty(a.shape, dtype=dtype, device=a.device)
    check(lib().sedd_cast(_p(a), _dt(a), _p(out), _dt(out), a.numel(), _stream()), "sedd_cast")
    return out


# ------------------------------------------------------------------------------------------------- GEMM
def gemm_nt(A, B, bias=None, gelu=False, out=None, out_dtype=None, accumulate=False, want_pre=False):
    """C[M,N] = A[M,K] @ B[N,K]^T (+bias)(GELU). A, B share dtype (bf16 or fp32), row strides allowed.
    -> C, or (C, pre) when gelu and want_pre; want_pre="grad": `pre` holds gelu'(pre-activation) instead of the
    pre-activation (for gemm_nn_dmul / gemm_nt_dmul)."""
    M, K = A.shape
    N, K2 = B.shape
    if K != K2 or A.dtype != B.dtype:
        raise ValueError(f"gemm_nt: A {tuple(A.shape)} {A.dtype} vs B {tuple(B.shape)} {B.dtype}")
    if A.stride(1) != 1 or B.stride(1) != 1:
        raise ValueError("gemm_nt: operands must be K-contiguous")
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or A.dtype, device=A.device)
    epi = _L.EPI_NONE if bias is None else (_L.EPI_BIAS_GELU if gelu else _L.EPI_BIAS)
    if gelu and want_pre == "grad":
        epi = _L.EPI_BIAS_GELU_GRAD
    if gelu and bias is None:
        raise ValueError("gemm_nt: GELU epilogue needs a bias")
    pre = torch.empty_like(out) if (gelu and want_pre) else None
    with _Timed("gemm_bf16_nt" if A.dtype == torch.bfloat16 else "gemm_f32_nt", 2.0 * M * N * K):
        if out.dtype == torch.float32 and not _ATOMIC and (accumulate or A.dtype == torch.float32):
            # launches that split K (small outputs, long reductions) fold their slices in a fixed order out of the
            # split-K scratch instead of adding to `out` with atomics
            ws = _tn_workspace(A.device)
            check(lib().sedd_gemm_nt_ws(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                        _p(bias), epi, _p(pre), 1 if accumulate else 0, M, N, K, _p(ws), ws.numel() * 4,
                                        _stream()), "sedd_gemm_nt_ws")
        else:
            check(lib().sedd_gemm_nt(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                     _p(bias), epi, _p(pre), 1 if accumulate else 0, M, N, K, _stream()), "sedd_gemm_nt")
    return (out, pre) if (gelu and want_pre) else out


# SEDD_ATOMIC_REDUCTIONS=1: no scratches for the cross-tile sums (fp32 atomics instead of fixed-order folds) — an A/B knob
_ATOMIC = bool(os.environ.get("SEDD_ATOMIC_REDUCTIONS"))


def _colsum_ws(like, M, N, colsum):
    """per-device fp32 scratch [ceil(M/256), N] for the dGELU epilogue's column sums (fixed-order fold instead of atomics);
    consecutive launches on a stream reuse it (the fold that reads it is ordered before the next launch)."""
    if colsum is None or _ATOMIC:
        return None
    return _scratch("colsum", like.device, ((M + 255) // 256) * N)


def gemm_nt_dgelu_supported(A, B, pre):
    if os.environ.get("SEDD_NO_DGELU"):
        return False
    M, K = A.shape
    N = B.shape[0]
    return (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and pre.dtype == torch.bfloat16 and M >= 192
            and N >= 96 and N % 8 == 0 and K % 32 == 0 and pre.is_contiguous() and pre.data_ptr() % 16 == 0
            and A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_nt_dgelu(A, B, pre, colsum=None):
    """C = (A[M,K] @ B[N,K]^T) * gelu_tanh'(pre[M,N]) (bf16); colsum fp32 [N] += C.sum(0) when given."""
    M, K = A.shape
    N = B.shape[0]
    if pre.shape != (M, N) or B.shape[1] != K:
        raise ValueError(f"gemm_nt_dgelu: A {tuple(A.shape)} B {tuple(B.shape)} pre {tuple(pre.shape)}")
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    with _Timed("gemm_bf16_nt", 2.0 * M * N * K):
        check(lib().sedd_gemm_nt_dgelu(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                       _p(pre), _p(colsum), _p(_colsum_ws(A, M, N, colsum)), M, N, K, _stream()),
              "sedd_gemm_nt_dgelu")
    return out


def set_gemm_cu_budget(cus):
    """CUs the persistent bf16 GEMM launches count on (multiple of 8, <= 256); process-wide (see sedd_gemm_set_cu_budget)."""
    check(lib().sedd_gemm_set_cu_budget(int(cus)), "sedd_gemm_set_cu_budget")


def gemm_dmul_shapes_ok(M, N, K, dtype):
    """whether the fused `dU = (dY W2) * gelu'` launch (sedd_gemm_nn_dmul / _nt_dmul) will take an [M, K] x [K, N] product
    of freshly allocated bf16 tensors — decided in the forward pass, which then stores gelu' instead of the pre-activation."""
    return (dtype == torch.bfloat16 and M >= 192 and N >= 96 and N % 8 == 0 and K % 32 == 0
            and not os.environ.get("SEDD_NO_DGELU") and not os.environ.get("SEDD_NO_DMUL"))


def gemm_dmul(A, B, mul, colsum=None, b_kn=True):
    """C = (A[M,K] @ B) * mul[M,N] (bf16), B [K, N] (b_kn) or [N, K]; colsum fp32 [N] += C.sum(0) when given."""
    M, K = A.shape
    N = B.shape[1] if b_kn else B.shape[0]
    if mul.shape != (M, N) or (B.shape[0] if b_kn else B.shape[1]) != K:
        raise ValueError(f"gemm_dmul: A {tuple(A.shape)} B {tuple(B.shape)} mul {tuple(mul.shape)}")
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    fn = lib().sedd_gemm_nn_dmul if b_kn else lib().sedd_gemm_nt_dmul
    with _Timed("gemm_bf16_nt", 2.0 * M * N * K):
        check(fn(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out), _p(mul), _p(colsum),
                 _p(_colsum_ws(A, M, N, colsum)), M, N, K, _stream()), "sedd_gemm_dmul")
    return out


def gemm_nt_rotary_supported(A, B, L, H, d):
    M, K = A.shape
    return (not os.environ.get("SEDD_NO_GEMM_ROTARY") and A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16
            and d == 64 and B.shape[0] == 3 * H * d and K % 32 == 0 and M >= 192 and M % L == 0
            and A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_nt_rotary(A, B, cos, sin, L, H, d, bias=None):
    """packed qkv [M, 3*H*d] = A[M,K] @ B[3*H*d, K]^T (+bias) with rotary applied to the q and k columns
    (position = row % L; cos / sin compact fp32 [L, d/2])."""
    M, K = A.shape
    N = B.shape[0]
    if cos.shape != (L, d // 2) or sin.shape != cos.shape or cos.dtype != torch.float32 or not cos.is_contiguous() \
            or not sin.is_contiguous():
        raise ValueError("gemm_nt_rotary: cos / sin must be contiguous fp32 [L, d/2]")
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    epi = _L.EPI_NONE if bias is None else _L.EPI_BIAS
    with _Timed("gemm_bf16_nt", 2.0 * M * N * K):
        check(lib().sedd_gemm_nt_rotary(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                        _p(bias), epi, _p(cos), _p(sin), L, H, d, M, N, K, _stream()),
              "sedd_gemm_nt_rotary")
    return out


def gemm_nn_supported(A, B):
    """whether sedd_gemm_nn / sedd_gemm_nn_dgelu take A [M, K] @ B [K, N] (else: transposed copy of B + gemm_nt)."""
    if os.environ.get("SEDD_NO_GEMM_NN"):
        return False
    M, K = A.shape
    N = B.shape[1]
    if A.dtype == torch.float32 and B.dtype == torch.float32:  # exact-fp32 MFMA kernel, any M
        return (N % 4 == 0 and K % 4 == 0 and A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 4 == 0
                and B.stride(0) % 4 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)
    return (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and M >= 192 and N >= 96 and N % 8 == 0
            and K % 32 == 0 and A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_nn(A, B, bias=None, out=None):
    """C[M,N] = A[M,K] @ B[K,N] (+bias), bf16 or fp32: the data gradient dY @ W against the weight as stored."""
    M, K = A.shape
    K2, N = B.shape
    if K != K2:
        raise ValueError(f"gemm_nn: A {tuple(A.shape)} vs B {tuple(B.shape)}")
    if out is None:
        out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    epi = _L.EPI_NONE if bias is None else _L.EPI_BIAS
    with _Timed("gemm_bf16_nt" if A.dtype == torch.bfloat16 else "gemm_f32_nt", 2.0 * M * N * K):
        if A.dtype == torch.float32 and not _ATOMIC:  # small fp32 outputs split K through the scratch
            ws = _tn_workspace(A.device)
            check(lib().sedd_gemm_nn_ws(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                        _p(bias), epi, M, N, K, _p(ws), ws.numel() * 4, _stream()), "sedd_gemm_nn_ws")
        else:
            check(lib().sedd_gemm_nn(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                     _p(bias), epi, M, N, K, _stream()), "sedd_gemm_nn")
    return out


def gemm_nn_dgelu(A, B, pre, colsum=None):
    """C = (A[M,K] @ B[K,N]) * gelu_tanh'(pre[M,N]) (bf16); colsum fp32 [N] += C.sum(0) when given."""
    M, K = A.shape
    N = B.shape[1]
    if pre.shape != (M, N) or B.shape[0] != K:
        raise ValueError(f"gemm_nn_dgelu: A {tuple(A.shape)} B {tuple(B.shape)} pre {tuple(pre.shape)}")
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    with _Timed("gemm_bf16_nt", 2.0 * M * N * K):
        check(lib().sedd_gemm_nn_dgelu(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                       _p(pre), _p(colsum), _p(_colsum_ws(A, M, N, colsum)), M, N, K, _stream()),
              "sedd_gemm_nn_dgelu")
    return out


def gemm_tn_supported(A, B):
    """whether sedd_gemm_tn takes these operands (else: transpose + gemm_nt)."""
    K, M = A.shape
    N = B.shape[1]
    if A.dtype == torch.float32 and B.dtype == torch.float32:  # exact-fp32 MFMA kernel, any K
        return (M % 4 == 0 and N % 4 == 0 and A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 4 == 0
                and B.stride(0) % 4 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)
    return (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and K % 32 == 0 and M % 8 == 0 and N % 8 == 0
            and A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def _tn_workspace(device):
    """split-K scratch of the weight-gradient / small-shape GEMMs: 256 partial tiles of 256 x 256 fp32 (64 MiB), fixed
    size, one per device and stream (see _scratch)."""
    return _scratch("splitk", device, 256 * 256 * 256)


def gemm_tn(A, B, out=None, out_dtype=None, accumulate=False, colsum=None):
    """C[M,N] (+)= A[K,M]^T @ B[K,N] (weight gradient: A = dY [tokens, out], B = X [tokens, in]); bf16 or fp32
    operands (fp32: fp32 C only).  colsum (bf16 operands, fp32 C): fp32 [M] that receives (+)= the column sums of A —
    the bias gradient — from the same launch, under the same `accumulate` flag as C."""
    K, M = A.shape
    K2, N = B.shape
    if K != K2 or A.dtype != B.dtype:
        raise ValueError(f"gemm_tn: A {tuple(A.shape)} {A.dtype} vs B {tuple(B.shape)} {B.dtype}")
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or torch.float32, device=A.device)
        accumulate = False
    ws = _tn_workspace(A.device) if out.dtype == torch.float32 else None
    if colsum is not None:
        if (A.dtype != torch.bfloat16 or out.dtype != torch.float32 or colsum.dtype != torch.float32
                or colsum.numel() != M or not colsum.is_contiguous()):
            raise ValueError("gemm_tn: colsum needs bf16 operands, an fp32 C and a contiguous fp32 [M]")
        with _Timed("gemm_bf16_nt", 2.0 * M * N * K):
            check(lib().sedd_gemm_tn_colsum(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _p(colsum),
                                            _dt(A), 1 if accumulate else 0, M, N, K, _p(ws), ws.numel() * 4, _stream()),
                  "sedd_gemm_tn_colsum")
        return out
    with _Timed("gemm_bf16_nt" if A.dtype == torch.bfloat16 else "gemm_f32_nt", 2.0 * M * N * K):
        check(lib().sedd_gemm_tn(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _dt(A), _dt(out),
                                 1 if accumulate else 0, M, N, K, _p(ws), 0 if ws is None else ws.numel() * 4,
                                 _stream()), "sedd_gemm_tn")
    return out


# --------------------------------------------------------------------------------------------- attention
def _attn_strides(t):
    """t: [B, L, H, d] view with contiguous last dim and head stride d -> (batch stride, seq stride)."""
    if t.stride(3) != 1 or t.stride(2) != t.shape[3]:
        raise ValueError("attention operands need layout [B, L, H, d] with contiguous (H, d)")
    return t.stride(0), t.stride(1)


def attn_fwd(q, k, v, key_mask=None, out=None):
    """q [B, Lq, H, d], k/v [B, Lk, H, d] (strided views of a packed qkv allowed; k and v share strides).
    key_mask: optional bool/uint8 [B, Lk], True = attend.  -> (o [B, Lq, H, d] contiguous, lse fp32 [B, H, Lq])."""
    B, Lq, H, d = q.shape
    Lk = k.shape[1]
    qs, ks = _attn_strides(q), _attn_strides(k)
    if _attn_strides(v) != ks:
        raise ValueError("k and v must share strides")
    o = torch.empty(B, Lq, H, d, dtype=q.dtype, device=q.device) if out is None else out
    _attn_strides(o)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
    km = None if key_mask is None else _c(key_mask.to(torch.uint8))
    with _Timed("attn_fwd", 4.0 * B * H * Lq * Lk * d):
        check(lib().sedd_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(km), _dt(q), B, H, Lq, Lk, d, qs[0], qs[1],
                                  ks[0], ks[1], o.stride(0), o.stride(1), _stream()), "sedd_attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o, dout, lse, dq, dk, dv, key_mask=None, rotary=None):
    """dq/dk/dv: preallocated views with the same strides as q/k/v (e.g. slots of a packed dqkv).
    rotary = (cos, sin) compact fp32 [L, d/2] tables: dq / dk come out as gradients of the PRE-rotary q / k (the
    conjugate rotation of rotary_qkv_(conj=True) applied as they are written)."""
    B, Lq, H, d = q.shape
    Lk = k.shape[1]
    qs, ks = _attn_strides(q), _attn_strides(k)
    if _attn_strides(dq) != qs or _attn_strides(dk) != ks or _attn_strides(dv) != ks or _attn_strides(v) != ks:
        raise ValueError("gradient views must share the strides of their primal tensors")
    if _attn_strides(dout) != _attn_strides(o):
        raise ValueError("dout must be laid out like o")
    delta = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
    km = None if key_mask is None else _c(key_mask.to(torch.uint8))
    rc, rs = rotary if rotary is not None else (None, None)
    if rc is not None and (rc.shape != (Lq, d // 2) or rs.shape != rc.shape or rc.dtype != torch.float32
                           or not rc.is_contiguous() or not rs.is_contiguous() or Lq != Lk):
        raise ValueError("attn_bwd: rotary tables must be contiguous fp32 [L, d/2] with Lq == Lk == L")
    with _Timed("attn_bwd", 10.0 * B * H * Lq * Lk * d):
        check(lib().sedd_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(km), _p(dq), _p(dk), _p(dv),
                                  _p(delta), _dt(q), B, H, Lq, Lk, d, qs[0], qs[1], ks[0], ks[1], o.stride(0),
                                  o.stride(1), _p(rc), _p(rs), _stream()), "sedd_attn_bwd")


def _packed_token_stride(t, name):
    """t: [B, L, H, d] view whose flattened (B L) token axis has ONE stride (batch stride = L x token stride)."""
    B, L, H, d = t.shape
    if t.stride(3) != 1 or t.stride(2) != d or (B > 1 and t.stride(0) != L * t.stride(1)):
        raise ValueError(f"{name}: need layout [B, L, H, d] with contiguous (H, d) and a uniform token stride")
    return t.stride(1)


def _check_cu(cu):
    if cu.dtype != torch.int32 or cu.dim() != 1 or cu.numel() < 1 or not cu.is_contiguous() or not cu.is_cuda:
        raise ValueError("cu_seqlens must be a contiguous int32 device vector (nseq + 1 entries, starting at 0)")


def attn_varlen_fwd(q, k, v, cu_seqlens, max_len):
    """flash_attn_varlen_qkvpacked_func's general form: q / k / v [B, L, H, d] views (slots of a packed qkv); the flattened
    (B L) token axis is cut into segments [cu[s], cu[s+1]); attention inside each segment.
    -> (o [B, L, H, d] contiguous, zero outside the segments; lse fp32 [H, B L])."""
    _cuda(q, k, v, cu_seqlens)
    _check_cu(cu_seqlens)
    B, L, H, d = q.shape
    T = B * L
    qs, ks = _packed_token_stride(q, "attn_varlen_fwd q"), _packed_token_stride(k, "attn_varlen_fwd k")
    if _packed_token_stride(v, "attn_varlen_fwd v") != ks:
        raise ValueError("k and v must share strides")
    o = torch.zeros(B, L, H, d, dtype=q.dtype, device=q.device)
    lse = torch.zeros(H, T, dtype=torch.float32, device=q.device)
    with _Timed("attn_fwd", 4.0 * T * H * max_len * d):
        check(lib().sedd_attn_varlen_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(cu_seqlens), cu_seqlens.numel() - 1, T,
                                         int(max_len), _dt(q), H, d, qs, ks, o.stride(1), _stream()), "sedd_attn_varlen_fwd")
    return o, lse


def attn_varlen_bwd(q, k, v, o, dout, lse, dq, dk, dv, cu_seqlens, max_len):
    """dq / dk / dv: preallocated (zeroed) views with the strides of q / k / v; gradients w.r.t. the q / k the kernel saw
    (post-rotary)."""
    _check_cu(cu_seqlens)
    B, L, H, d = q.shape
    T = B * L
    qs, ks = _packed_token_stride(q, "attn_varlen_bwd q"), _packed_token_stride(k, "attn_varlen_bwd k")
    if (_packed_token_stride(dq, "dq") != qs or _packed_token_stride(dk, "dk") != ks or _packed_token_stride(dv, "dv") != ks
            or _packed_token_stride(v, "v") != ks):
        raise ValueError("gradient views must share the strides of their primal tensors")
    os_ = _packed_token_stride(o, "o")
    if _packed_token_stride(dout, "dout") != os_:
        raise ValueError("dout must be laid out like o")
    delta = torch.zeros(H, T, dtype=torch.float32, device=q.device)
    with _Timed("attn_bwd", 10.0 * T * H * max_len * d):
        check(lib().sedd_attn_varlen_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(dq), _p(dk), _p(dv), _p(delta),
                                         _p(cu_seqlens), cu_seqlens.numel() - 1, T, int(max_len), _dt(q), H, d, qs, ks,
                                         os_, _stream()), "sedd_attn_varlen_bwd")


# ------------------------------------------------------------------------------------------ optimiser
def sumsq_clip(g, gscale=1.0, max_norm=0.0, ws=None, out=None):
    if ws is None:
        ws = torch.empty(1024, dtype=torch.float32, device=g.device)
    if out is None:
        out = torch.ones(2, dtype=torch.float32, device=g.device)
    check(lib().sedd_sumsq_clip(_p(g), g.numel(), float(gscale), float(max_norm), _p(ws), _p(out), _stream()),
          "sedd_sumsq_clip")
    return out


def clip_from_sumsq(out, gscale, max_norm):
    check(lib().sedd_clip_from_sumsq(_p(out), float(gscale), float(max_norm), _stream()), "sedd_clip_from_sumsq")
    return out


def adamw_ema_step(p, g, m, v, ema, p_bf16, clip_coef, grad_scale, lr, beta1, beta2, eps, weight_decay, step, ema_decay,
                   decoupled=True):
    """decoupled=True: AdamW (p *= 1 - lr*wd); False: torch.optim.Adam's L2 form (g += wd*p before the moments)."""
    check(lib().sedd_adamw_ema_step(_p(p), _p(g), _p(m), _p(v), _p(ema), _p(p_bf16), _p(clip_coef), float(grad_scale),
                                    float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                    1 if decoupled else 0, int(step), float(ema_decay), p.numel(), _stream()),
          "sedd_adamw_ema_step")


def adamw_ema_double_step(p, g, m, v, ema, p_bf16, clip_coef, grad_scale, lr, beta1, beta2, eps, weight_decay, step,
                          ema_decay, decoupled=True):
    """steps `step` and `step + 1` on the same gradients, EMA between them (the dual-track trainer's quirk Q2), one pass."""
    check(lib().sedd_adamw_ema_double_step(_p(p), _p(g), _p(m), _p(v), _p(ema), _p(p_bf16), _p(clip_coef),
                                           float(grad_scale), float(lr), float(beta1), float(beta2), float(eps),
                                           float(weight_decay), 1 if decoupled else 0, int(step), float(ema_decay),
                                           p.numel(), _stream()), "sedd_adamw_ema_double_step")


def adamw_step_scalars(lr, beta1, beta2, weight_decay, decoupled, step, ema_decay):
    """the six per-step scalars of the optimiser kernel (host arithmetic of the library, so the captured and the eager
    step use the very same numbers)."""
    import ctypes
    out = (ctypes.c_float * 6)()
    check(lib().sedd_adamw_step_scalars(float(lr), float(beta1), float(beta2), float(weight_decay), 1 if decoupled else 0,
                                        int(step), float(ema_decay), ctypes.addressof(out)), "sedd_adamw_step_scalars")
    return [float(x) for x in out]


def write_step_scalars(hyper_dev, salt_dev, scalars, salt):
    """hyper_dev[0..5] <- scalars, salt_dev[0] <- salt, in stream order (values travel as launch arguments)."""
    salt = int(salt) & 0x7FFFFFFFFFFFFFFF
    check(lib().sedd_write_step_scalars(_p(hyper_dev), _p(salt_dev), *[float(x) for x in scalars], salt, _stream()),
          "sedd_write_step_scalars")


def adamw_ema_step_dev(p, g, m, v, ema, p_bf16, clip_coef, hyper, grad_scale, beta1, beta2, eps, weight_decay, steps=1,
                       decoupled=True):
    """sedd_adamw_ema_step / _double_step (steps = 1 | 2) with lr / bias corrections / EMA decay read from the device
    buffer `hyper` (see write_step_scalars): the form a HIP-graph capture of the train step launches."""
    check(lib().sedd_adamw_ema_step_dev(_p(p), _p(g), _p(m), _p(v), _p(ema), _p(p_bf16), _p(clip_coef), _p(hyper),
                                        float(grad_scale), float(beta1), float(beta2), float(eps), float(weight_decay),
                                        1 if decoupled else 0, int(steps), p.numel(), _stream()), "sedd_adamw_ema_step_dev")


def ema_update_(shadow, p, one_minus_decay):
    check(lib().sedd_ema_update(_p(shadow), _p(p), float(one_minus_decay), shadow.numel(), _stream()), "sedd_ema_update")
    return shadow


def scale_(x, s):
    """x *= s in place, s a 1-element fp32 device tensor (never read on the host)."""
    if s.dtype != torch.float32 or s.numel() != 1 or not x.is_contiguous():
        raise ValueError("scale_: contiguous x and a 1-element fp32 scale")
    check(lib().sedd_scale_(_p(x), _dt(x), _p(s), x.numel(), _stream()), "sedd_scale_")
    return x


def batch_weighted_sum(x, w):
    """x fp32 [B, L], w fp32 [B] -> 0-dim fp32 tensor sum_b w_b * sum_l x[b, l]."""
    B = x.shape[0]
    M = x.numel()
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    check(lib().sedd_batch_weighted_sum(_p(_c(x, torch.float32)), _p(_batch_scalar(w, B)), M, M // B, _p(ws), _p(out),
                                        _stream()), "sedd_batch_weighted_sum")
    return out[0]
