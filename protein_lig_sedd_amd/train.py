"""Data-parallel training step for the SEDD hot path — the MI355X-native replacement of the reference's `config_ddp`
path (training/run_train_ddp_protlig_transf.py:160-165 NCCL init, :196-197 DDP wrap, :284-285 step + loss all-reduce;
processing/losses.py:49-117 warm-up / clip / AdamW / EMA).

Design (one process per GPU, torch.distributed backend "nccl" == RCCL over xGMI):
  * all parameters live in ONE flat fp32 buffer (each tensor padded to 64 elements); gradients, Adam moments, the EMA
    shadow and the bf16 compute copy are flat buffers of the same layout.  param.data / param.grad are views, so the
    backward kernels accumulate weight gradients straight into the flat gradient buffer;
  * the gradient buffer is all-reduced in ONE RCCL call after backward (default), or — opt-in, SEDD_DP_OVERLAP=1 — in
    per-block buckets launched asynchronously as soon as a block's backward has finished (see SEDDTrainer.__init__
    for why that is not the default with persistent one-block-per-CU GEMMs); then one deterministic norm reduction, then
    ONE fused kernel does clip -> AdamW -> EMA -> bf16 weight copy over the whole model (sedd_adamw_ema_step) instead
    of the reference's per-parameter optimizer / EMA loops;
  * no DDP wrapper, no GradScaler (bf16 needs no loss scaling; the reference's scaler is a no-op numerically).
"""
import os

import torch
import torch.distributed as dist

from . import ops
from .losses import get_loss_fn
from .model import layers as Lyr
from .model.ema import ExponentialMovingAverage
from .model.transformer import set_training

ALIGN = 64


class FlatState:
    """Flat parameter / gradient / optimiser-state buffers for a list of parameters."""

    def __init__(self, params, compute_dtype=torch.bfloat16, with_ema=True, multiple=ALIGN):
        params = list(params)
        self.params = [p for p in params if p.requires_grad]
        # position of every trainable parameter in the full parameter list: torch optimizers index their state by it
        # (frozen parameters keep their slot and simply have no state), so reference checkpoints of task-frozen runs map
        # onto the right tensors
        self.index = [i for i, p in enumerate(params) if p.requires_grad]
        self.n_all = len(params)
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        n = (n + multiple - 1) // multiple * multiple  # (sharded optimiser: equal 64-aligned shards, zero tail)
        self.n = n
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.compute_dtype = compute_dtype
        self.pc = torch.zeros(n, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        for p, off in zip(self.params, self.offsets):
            view = self.p[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.g[off:off + p.numel()].view(p.shape)
        self.ema = self.p.clone() if with_ema else None
        if self.pc is not None:
            ops.cast(self.p, torch.bfloat16, out=self.pc)
        self.refresh_compute_weights()

    def view(self, flat, i):
        p, off = self.params[i], self.offsets[i]
        return flat[off:off + p.numel()].view(p.shape)

    def refresh_compute_weights(self):
        """(re)attach the compute-dtype copies after the flat buffers changed behind autograd's back."""
        views = getattr(self, "_cw_views", None)
        if views is None:  # the views never change, only the parameter versions they are tagged with
            views = self._cw_views = [(p, p.detach() if self.pc is None else self.view(self.pc, i))
                                      for i, p in enumerate(self.params) if p.dim() == 2]
        for p, v in views:
            Lyr.set_compute_weight(p, self.compute_dtype, v)

    def ema_params(self):
        return [self.view(self.ema, i) for i in range(len(self.params))]


def fl_is_cuda(flat):
    return flat.p.is_cuda


def init_distributed():
    """Bootstrap from torch.distributed.run's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), as
    run_train_ddp_protlig_transf.py:160-163 does; backend "nccl" is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SEDD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        if backend != "nccl":
            # gloo stages device tensors through the host from its own threads; on this stack every such all-reduce of
            # a buffer that the launch stream is still writing hung (round-1 rehearsals, also with plain torch ops and
            # none of this package's kernels).  The collectives of this path are RCCL's; gloo is for CPU tensors.
            raise RuntimeError(f"init_distributed: backend {backend!r} cannot carry GPU tensors here — use 'nccl' "
                               "(RCCL); gloo is supported for CPU tensors only")
        torch.cuda.set_device(local)
    if (world > 1 or os.environ.get("SEDD_DP_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


class SEDDTrainer:
    """loss_fn(model, batch) -> backward -> RCCL all-reduce of the flat gradient buffer -> fused clip/AdamW/EMA step.

    optim: namespace/dict with lr, beta1, beta2, eps, weight_decay, warmup, grad_clip (configs/*.yaml `optim:`)."""

    def __init__(self, model, noise, graph, optim, ema_decay=0.9999, sampling_eps=1e-3, world_size=None,
                 process_group=None, overlap=True, lr_schedule=None, ema=None, conditioners=False, accum=1,
                 hip_graph=False, grad_comm_dtype=None, zero1=None, wgrad_stream=None):
        g = (lambda k, d=None: optim.get(k, d)) if isinstance(optim, dict) else (lambda k, d=None: getattr(optim, k, d))
        self.lr, self.beta1, self.beta2 = g("lr", 3e-4), g("beta1", 0.9), g("beta2", 0.999)
        self.eps, self.wd = g("eps", 1e-8), g("weight_decay", 0.0)
        self.warmup, self.grad_clip = g("warmup", 0), g("grad_clip", 1.0)
        # losses.py:35-46: 'Adam' puts weight_decay into the gradient (L2), 'AdamW' decouples it; the fused kernel takes
        # either form.  Anything else is the reference's NotImplementedError.
        self.optimizer_name = g("optimizer", "AdamW")
        if self.optimizer_name not in ("Adam", "AdamW"):
            raise NotImplementedError(f'Optimizer {self.optimizer_name} not supported yet!')
        self.decoupled_wd = self.optimizer_name == "AdamW"
        if g("accum", 1) not in (None, 1):
            raise NotImplementedError("gradient accumulation through the optimiser config (training.accum != 1): the "
                                      "reference's step_fn resets its counter every call (losses.py:84) and only works "
                                      "for accum == 1; pass accum=k to SEDDTrainer for real accumulation (opt-in)")
        # accum = k > 1 (opt-in; what losses.py:80-115 was written to do before its counter reset broke it): k calls of
        # train_step each add loss / k's gradients to the flat buffer — it is NOT zeroed in between — and the k-th call
        # all-reduces once, steps the optimiser and returns the summed loss (the mean over the k micro-batches)
        self.accum = int(accum)
        if self.accum < 1:
            raise ValueError("accum must be >= 1")
        self._accum_iter, self._accum_loss = 0, None
        # hip_graph=True (opt-in, not in the reference): train_step runs as one replayed HIP graph, see _graphed_step
        self.hip_graph, self.graph_warmup = bool(hip_graph), 2
        self._graphs, self._graph_calls, self._hyper, self._salt, self._salt_count = {}, {}, None, None, 0
        self._capturing, self._always_norm = False, False
        self.model, self.noise, self.graph = model, noise, graph
        # optional step -> lr callable (e.g. lr_scheduler.warmup_cosine_lr of the cross-attention trainer,
        # run_train_protlig_cross_att.py); default = losses.py:58-62 linear warm-up
        self.lr_schedule = lr_schedule
        self.pg = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        cdt = getattr(model, "compute_dtype", torch.float32)
        # Two ways to make the per-step exchange cheaper than one fp32 all-reduce of the flat gradient buffer (8 P (N-1)/N
        # bytes per GPU), both opt-in (arguments, or SEDD_DP_GRAD_DTYPE=bf16 / SEDD_DP_ZERO1=1):
        #   grad_comm_dtype=torch.bfloat16: the gradients travel as bf16 (cast -> collective -> cast back; half the bytes;
        #       the sum over ranks is then rounded per hop — the usual compressed-gradient trade);
        #   zero1=True: reduce-scatter -> clip / Adam(W) / EMA on this rank's 1/N shard of the flat buffers -> all-gather of
        #       the updated fp32 parameters (+ a local bf16 cast): same bytes as the all-reduce in fp32 (6 P (N-1)/N with
        #       bf16 gradients), and the 38 B/param optimiser pass shrinks N-fold (cfg 3: 4.4 ms -> 0.55 ms at N = 8).
        #       Adam moments and the EMA shadow are current on the owning rank only; sync_sharded_state() gathers them
        #       (called by state_dict / ema.copy_to / ema.state_dict).
        if grad_comm_dtype is None:
            grad_comm_dtype = torch.bfloat16 if os.environ.get("SEDD_DP_GRAD_DTYPE", "") == "bf16" else torch.float32
        if grad_comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_comm_dtype must be torch.float32 or torch.bfloat16")
        self.grad_comm_dtype = grad_comm_dtype
        self.zero1 = bool(os.environ.get("SEDD_DP_ZERO1") == "1" if zero1 is None else zero1)
        nshard = max(self.world, 1) if self.zero1 else 1
        self.flat = FlatState(list(model.parameters()), cdt, with_ema=True, multiple=ALIGN * nshard)
        self.ema_decay = ema_decay
        if ema is None:
            self.ema = ExponentialMovingAverage([], ema_decay)
        else:  # adopt an EMA object the caller already holds (losses.get_step_fn): its shadow moves into the flat buffer
            self.ema = ema
            for dst, src in zip(self.flat.ema_params(), ema.shadow_params):
                dst.copy_(src)
        self.ema.shadow_params = self.flat.ema_params()
        # conditioners=True: train_step(..., esm_cond=, mol_cond=) reaches SEDD.forward's pooled conditioners (the
        # reference's loss_fn drops them, losses.py:26 — default)
        self.loss_fn = get_loss_fn(noise, graph, train=True, sampling_eps=sampling_eps, conditioners=conditioners)
        self.step_count = 0
        dev = self.flat.p.device
        self._ws = torch.empty(1024, dtype=torch.float32, device=dev)
        self._norm = torch.ones(2, dtype=torch.float32, device=dev)
        # SEDD_DP_FORCE=1 runs every collective of the N > 1 path over a 1-rank group (rehearsal of the RCCL stream
        # ordering on a single GPU); SEDD_DP_OVERLAP=0 falls back to one blocking all-reduce after backward
        self.dp = self.world > 1 or (os.environ.get("SEDD_DP_FORCE") == "1" and dist.is_initialized())
        # The per-block overlapped all-reduce is OPT-IN (SEDD_DP_OVERLAP=1).  The GEMMs of this path are persistent, one
        # 512-thread block per CU holding the CU's whole register file; a CU that hosts an RCCL workgroup cannot take a
        # GEMM block, so every GEMM launched while a collective is resident waits for a second round of its blocks
        # (2x its time).  One all-reduce of the flat gradient buffer after backward (~1.9 ms at 8 GPUs for 370 MB)
        # costs less than that and nothing at all at N = 1.
        self.overlap = overlap and self.dp and os.environ.get("SEDD_DP_OVERLAP", "0") == "1"
        self.zero1 = self.zero1 and self.dp
        if self.overlap and fl_is_cuda(self.flat):
            # leave the collective's workgroups a CU per XCD: the one-block-per-CU GEMMs then still fit one round
            ops.set_gemm_cu_budget(int(os.environ.get("SEDD_GEMM_CUS", "248")))
        if self.overlap and (self.zero1 or self.grad_comm_dtype != torch.float32):
            raise ValueError("SEDD_DP_OVERLAP=1 (per-block buckets) excludes zero1 / bf16 gradient exchange")
        self.rank = dist.get_rank(process_group) if (self.dp and dist.is_initialized()) else 0
        per = self.flat.n // (self.world if self.zero1 else 1)
        self.shard = (self.rank * per, (self.rank + 1) * per) if self.zero1 else (0, self.flat.n)
        # wgrad_stream: weight- and bias-gradient launches of backward go to a second stream (ops.SideStream) and run
        # beside the data-gradient chain — for steps whose kernels each fill a fraction of the chip.  True / False, or None
        # (default) = where it was measured to pay on every box: fp32 compute and steps of at most `_side_max` tokens
        # (8192).  BASELINE cfg 1 (1024 tokens), same-box A/B x3: fp32 7.94 -> 7.19 ms; bf16 5.2 -> 5.55 ms — that step is
        # bound by the host's enqueue rate, which the forks add to (on a faster host the same A/B read 5.3 -> 4.9 ms), so
        # bf16 stays opt-in here; at cfg 2's 32768 tokens every GEMM fills the chip and it buys 0.5 %.
        if os.environ.get("SEDD_WGRAD_STREAM") in ("0", "1"):  # (A/B override)
            wgrad_stream = os.environ["SEDD_WGRAD_STREAM"] == "1"
        if wgrad_stream is None and getattr(model, "compute_dtype", None) != torch.float32:
            wgrad_stream = False
        if self.overlap:  # the per-block buckets are launched from the backward hooks: a block's weight gradients must be
            wgrad_stream = False  # in the gradient buffer by then, not waiting in the side stream's queue
        self._side = ops.SideStream(self.flat.p.device) if (wgrad_stream is not False and self.flat.p.is_cuda) else None
        self._side_max = int(os.environ.get("SEDD_WGRAD_STREAM_MAX_TOKENS", "8192")) if wgrad_stream is None else None
        self._gb = None  # bf16 staging buffer of the gradient exchange
        self._shards_dirty = False
        if self.zero1:
            self.ema._before_read = self.sync_sharded_state
        self._pending = []
        self._ar_events = []
        self._buckets = self._make_buckets()
        if self.dp:
            self.broadcast_parameters()

    # ---- gradient buckets: [embedding + sigma_map], one per block, [output layer] -------------------------------
    def _make_buckets(self):
        ids = {id(p): i for i, p in enumerate(self.flat.params)}
        buckets, assigned = [], set()
        blocks = list(getattr(self.model, "blocks", []))

        def rng(ps):
            idx = sorted(ids[id(p)] for p in ps if id(p) in ids)
            if not idx:
                return None
            assigned.update(idx)
            lo = self.flat.offsets[idx[0]]
            last = idx[-1]
            hi = self.flat.offsets[last + 1] if last + 1 < len(self.flat.offsets) else self.flat.n
            return (lo, hi)

        for b, blk in enumerate(blocks):
            r = rng(list(blk.parameters()))
            if r:
                buckets.append(r)
                if self.overlap:
                    blk._on_grads_ready = (lambda rr: (lambda: self._launch_allreduce(rr)))(r)
        self._rest = [i for i in range(len(self.flat.params)) if i not in assigned]
        return buckets

    def _launch_allreduce(self, r):
        self._pending.append(dist.all_reduce(self.flat.g[r[0]:r[1]], group=self.pg, async_op=True))

    def broadcast_parameters(self, src=0):
        dist.broadcast(self.flat.p, src, group=self.pg)
        # the EMA shadow travels as its own buffer: on a fresh start it equals rank src's parameters anyway, on a resume
        # (losses.get_step_fn adopting a restored state['ema']) it is the restored shadow and must survive the binding
        dist.broadcast(self.flat.ema, src, group=self.pg)
        if self.flat.pc is not None:
            ops.cast(self.flat.p, torch.bfloat16, out=self.flat.pc)
        self.flat.refresh_compute_weights()

    def current_lr(self, step):
        sched = getattr(self, "lr_schedule", None)
        if sched is not None:
            return float(sched(step))
        return self.lr * min(step / self.warmup, 1.0) if self.warmup and self.warmup > 0 else self.lr

    def train_step(self, batch, t=None, perturbed_batch=None, u=None, esm_cond=None, mol_cond=None):
        """One call of losses.py:80-115's step_fn.  accum == 1: one optimisation step, returns the local loss (0-dim
        device tensor).  accum = k: a micro-step — the k-th call steps the optimiser and returns the accumulated loss,
        the others return their own loss / k, as the reference's step_fn does."""
        fl = self.flat
        if self.hip_graph and self.accum == 1:
            inputs = {k: v for k, v in dict(batch=batch, t=t, perturbed_batch=perturbed_batch, u=u, esm_cond=esm_cond,
                                            mol_cond=mol_cond).items() if v is not None}

            def body(batch, t=None, perturbed_batch=None, u=None, esm_cond=None, mol_cond=None):
                fl.g.zero_()
                set_training(self.model, True)
                loss = self.loss_fn(self.model, batch, t=t, perturbed_batch=perturbed_batch, u=u, esm_cond=esm_cond,
                                    mol_cond=mol_cond)
                self._backward(loss, batch.numel())
                return (loss,)

            return self._graphed_step(body, inputs)[0]
        if self._accum_iter == 0:
            fl.g.zero_()
        set_training(self.model, True)
        loss = self.loss_fn(self.model, batch, t=t, perturbed_batch=perturbed_batch, u=u, esm_cond=esm_cond,
                            mol_cond=mol_cond)
        if self.accum == 1:
            self._backward(loss, batch.numel())
            self.apply_gradients()
            return loss.detach()
        if self.overlap:
            raise RuntimeError("gradient accumulation with SEDD_DP_OVERLAP=1: the per-block buckets would all-reduce "
                               "every micro-step; use the default single all-reduce")
        loss = loss / self.accum
        self._backward(loss, batch.numel())   # every backward kernel ACCUMULATES into the flat gradient buffer
        self._accum_iter += 1
        self._accum_loss = loss.detach() if self._accum_loss is None else self._accum_loss + loss.detach()
        if self._accum_iter < self.accum:
            return loss.detach()
        total, self._accum_iter, self._accum_loss = self._accum_loss, 0, None
        self.apply_gradients()
        return total

    def apply_gradients(self, step=None):
        """all-reduce -> clip -> Adam(W) -> EMA on whatever the flat gradient buffer holds (losses.py:52-69 + :102).
        step: the 1-based optimiser step this is (default: the next one)."""
        self._reduce_gradients()
        self._launch_update(*self._host_advance(step))
        self.flat.refresh_compute_weights()

    # ---- the optimiser pass in two halves: host bookkeeping (what changes from step to step) and the launches ----------
    def _host_advance(self, step=None):
        """-> (lr, ema decay, optimiser step, kernel steps): pure host arithmetic, no launches."""
        self.step_count = self.step_count + 1 if step is None else int(step)
        self.ema.num_updates += 1
        decay = min(self.ema_decay, (1 + self.ema.num_updates) / (10 + self.ema.num_updates))
        return self.current_lr(self.step_count), decay, self.step_count, 1

    def _launch_update(self, lr, decay, opt_step, ns):
        """clip -> Adam(W) -> EMA -> bf16 copy.  Inside a captured step (self._hyper set) the per-step scalars come from
        device memory, so the frozen launch arguments stay valid for every replay."""
        fl = self.flat
        gscale = 1.0 / self.world
        clip = None
        if self.zero1:
            # this rank's shard only: partial sum of squares -> all-reduce of ONE float -> the clip coefficient of the
            # whole model -> clip / Adam(W) / EMA on the shard -> all-gather of the fp32 parameters -> local bf16 copy
            lo, hi = self.shard
            clipping = self.grad_clip is not None and self.grad_clip >= 0
            if clipping or self._always_norm:
                ops.sumsq_clip(fl.g[lo:hi], gscale, -1.0, self._ws, self._norm)
                dist.all_reduce(self._norm[:1], group=self.pg)
                if clipping:
                    ops.clip_from_sumsq(self._norm, gscale, self.grad_clip)
                    clip = self._norm[1:]
            step_fn = ops.adamw_ema_double_step if ns == 2 else ops.adamw_ema_step
            step_fn(fl.p[lo:hi], fl.g[lo:hi], fl.m[lo:hi], fl.v[lo:hi], fl.ema[lo:hi], None, clip, gscale, lr, self.beta1,
                    self.beta2, self.eps, self.wd, opt_step, decay, self.decoupled_wd)
            self._gather_flat(fl.p)
            if fl.pc is not None:
                ops.cast(fl.p, torch.bfloat16, out=fl.pc)
            self._shards_dirty = True
            return
        if self.grad_clip is not None and self.grad_clip >= 0:
            ops.sumsq_clip(fl.g, gscale, self.grad_clip, self._ws, self._norm)
            clip = self._norm[1:]
        elif self._always_norm:
            ops.sumsq_clip(fl.g, gscale, -1.0, self._ws, self._norm)
        if self._capturing:
            ops.adamw_ema_step_dev(fl.p, fl.g, fl.m, fl.v, fl.ema, fl.pc, clip, self._hyper, gscale, self.beta1, self.beta2,
                                   self.eps, self.wd, ns, self.decoupled_wd)
        elif ns == 2:
            ops.adamw_ema_double_step(fl.p, fl.g, fl.m, fl.v, fl.ema, fl.pc, clip, gscale, lr, self.beta1, self.beta2,
                                      self.eps, self.wd, opt_step, decay, self.decoupled_wd)
        else:
            ops.adamw_ema_step(fl.p, fl.g, fl.m, fl.v, fl.ema, fl.pc, clip, gscale, lr, self.beta1, self.beta2, self.eps,
                               self.wd, opt_step, decay, self.decoupled_wd)

    # ---- whole-step HIP graph ------------------------------------------------------------------------------------------
    def _graphed_step(self, body, inputs):
        """Run `body(**inputs)` (zero grads -> loss -> backward; returns a tuple of loss tensors) + the optimiser pass as ONE
        replayed HIP graph.  A small-batch step is thousands of launches of a few microseconds and the Python between them
        is the limiter (cfg 3 at batch 4: ~36 ms of host time per 41 ms step); a replay is one launch.  What changes from
        step to step lives in device memory: the inputs (static buffers, copied per call), lr / bias corrections / EMA
        decay (`_hyper`, sedd_adamw_ema_step_dev) and the dropout salt (indirect seeds, ops.DropSeed) — both written by
        sedd_write_step_scalars right before the replay; t / corruption noise come from torch's graph-safe generator.
        The first `graph_warmup` calls run eagerly (allocator pools, scratches and weight caches settle), the next one
        captures.  Single process only (the RCCL all-reduce stays outside captures here)."""
        if self.dp:
            raise RuntimeError("hip_graph=True: single-process training only")
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(inputs.items()))
        ent = self._graphs.get(key)
        if ent is None:
            n = self._graph_calls.get(key, 0)
            self._graph_calls[key] = n + 1
            if n < self.graph_warmup:
                out = body(**inputs)
                self._launch_update(*self._host_advance())
                self.flat.refresh_compute_weights()
                return tuple(o.detach() for o in out)
            dev = self.flat.p.device
            if self._hyper is None:
                self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)
                self._salt = torch.zeros(1, dtype=torch.int64, device=dev)
            static = {k: v.clone() for k, v in inputs.items()}
            graph = torch.cuda.CUDAGraph()
            ops.DropSeed._indirect = (self._salt, 0)
            self._capturing = True
            try:
                with torch.cuda.graph(graph):
                    out = body(**static)
                    self._launch_update(0.0, 0.0, 1, self._kernel_steps())
            except RuntimeError as e:
                graph = None
                raise RuntimeError(f"hip_graph=True: the train step could not be captured ({e})") from e
            finally:
                ops.DropSeed._indirect = None
                self._capturing = False
            ent = self._graphs[key] = (graph, static, tuple(o.detach() for o in out))
        graph, static, out = ent
        for k, v in inputs.items():
            static[k].copy_(v)
        lr, decay, opt_step, ns = self._host_advance()
        self._salt_count += 1
        ops.write_step_scalars(self._hyper, self._salt,
                               ops.adamw_step_scalars(lr, self.beta1, self.beta2, self.wd, self.decoupled_wd, opt_step, decay),
                               torch.initial_seed() * 0x9E3779B97F4A7C15 + self._salt_count * 0xA24BAED4963EE407)
        graph.replay()
        return tuple(o.clone() for o in out)

    def _kernel_steps(self):
        return 1

    def reset_graphs(self):
        """drop the captured steps (after anything a capture froze has changed: shapes are keyed, module structure,
        trainable set and hyper-parameters other than lr / step / EMA decay are not)."""
        self._graphs, self._graph_calls = {}, {}

    def _backward(self, loss, tokens=None):
        """loss.backward() with the weight-gradient launches on the side stream when one is configured (and, in the
        by-size mode, the step has at most `_side_max` tokens); on return every gradient launch is ordered before whatever
        the current stream does next."""
        if self._side is None or (self._side_max is not None and (tokens is None or tokens > self._side_max)):
            loss.backward()
            return
        ops.SIDE = self._side
        try:
            loss.backward()
        finally:
            ops.SIDE = None
            self._side.join()

    def _reduce_gradients(self):
        """Finish the gradient all-reduce (SUM; the 1/world averaging is folded into the optimiser kernel's grad_scale):
        block buckets were launched asynchronously from the backward hooks, the leftovers (embedding, sigma_map, output
        layer) go now; then wait for everything."""
        if not getattr(self, "dp", self.world > 1):
            return
        fl = self.flat
        if self.overlap:
            for lo, hi in self._leftover_ranges({(r[0], r[1]) for r in self._buckets}):
                self._pending.append(dist.all_reduce(fl.g[lo:hi], group=self.pg, async_op=True))
            for w in self._pending:
                w.wait()
            self._pending = []
        else:
            timed = fl.g.is_cuda and len(self._ar_events) < 256
            if timed:  # the launch stream waits for RCCL's stream here, so the pair brackets the collective
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._exchange_gradients()
            if timed:
                e1.record()
                self._ar_events.append((e0, e1))

    def _exchange_gradients(self):
        """SUM over ranks of the flat gradient buffer, into fl.g (whole buffer) or — zero1 — into this rank's shard of it."""
        fl = self.flat
        lo, hi = self.shard
        buf = fl.g
        if self.grad_comm_dtype == torch.bfloat16:
            if self._gb is None:
                self._gb = torch.empty(fl.n, dtype=torch.bfloat16, device=fl.g.device)
            buf = self._gb
            if fl.g.is_cuda:
                ops.cast(fl.g, torch.bfloat16, out=buf)
            else:
                buf.copy_(fl.g)  # (CPU tensors only exist in the gloo plumbing tests)
        if self.zero1:
            self._reduce_scatter(buf[lo:hi], buf)
        else:
            dist.all_reduce(buf, group=self.pg)
        if buf is not fl.g:
            if fl.g.is_cuda:
                ops.cast(buf[lo:hi], torch.float32, out=fl.g[lo:hi])
            else:
                fl.g[lo:hi].copy_(buf[lo:hi])

    def _reduce_scatter(self, out, buf):
        """out (= this rank's slice of buf, in place) <- SUM over ranks of that slice.  RCCL: one reduce-scatter; gloo (CPU
        plumbing tests) has no reduce-scatter, so it all-reduces and keeps the slice."""
        if dist.get_backend(self.pg) == "gloo":
            dist.all_reduce(buf, group=self.pg)
        else:
            dist.reduce_scatter_tensor(out, buf, group=self.pg)

    def _gather_flat(self, flat):
        """all-gather of a flat buffer whose shard [lo, hi) is current on each rank, in place."""
        lo, hi = self.shard
        if dist.get_backend(self.pg) == "gloo":
            parts = [torch.empty_like(flat[lo:hi]) for _ in range(self.world)]
            dist.all_gather(parts, flat[lo:hi].contiguous(), group=self.pg)
            for r, part in enumerate(parts):
                flat[r * (hi - lo):(r + 1) * (hi - lo)].copy_(part)
        else:
            dist.all_gather_into_tensor(flat, flat[lo:hi], group=self.pg)

    def sync_sharded_state(self):
        """zero1: bring the Adam moments and the EMA shadow of every shard to this rank (checkpoints, EMA evaluation)."""
        if self.zero1 and self._shards_dirty:
            for flat in (self.flat.m, self.flat.v, self.flat.ema):
                self._gather_flat(flat)
            self._shards_dirty = False

    def reset_allreduce_timing(self):
        self._ar_events = []

    def allreduce_ms_per_step(self):
        """mean time of the (non-overlapped) gradient all-reduce since the last reset, from HIP events on the launch
        stream; None when no collective ran (single process) or in the overlapped form."""
        if not self._ar_events:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._ar_events) / len(self._ar_events)

    def _leftover_ranges(self, done):
        cov = sorted(done)
        out, pos = [], 0
        for lo, hi in cov:
            if lo > pos:
                out.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.flat.n:
            out.append((pos, self.flat.n))
        return out

    def grad_norm(self):
        """global gradient norm of the last step (after averaging over ranks)."""
        return float(self._norm[0]) ** 0.5 / self.world

    def reduce_loss(self, loss):
        """scalar loss all-reduce(AVG) for logging (run_train_ddp_protlig_transf.py:285)."""
        if self.world > 1:
            loss = loss.clone()
            dist.all_reduce(loss, group=self.pg)
            loss /= self.world
        return loss

    # ---- checkpoint in the reference's layout (utils/utils.py:73-95: optimizer / model / ema / step) -------------
    def optimizer_state_dict(self):
        """torch.optim.Adam/AdamW `state_dict()` layout (what utils/utils.py:88 stores and :80 loads): per-parameter
        `step` / `exp_avg` / `exp_avg_sq` in `model.parameters()` order, views of the flat moment buffers."""
        fl = self.flat
        self.sync_sharded_state()
        step = torch.tensor(float(self._adam_steps()))
        # keys = positions in model.parameters() (frozen parameters own a slot but no state, as in torch)
        state = {fl.index[i]: {"step": step.clone(), "exp_avg": fl.view(fl.m, i), "exp_avg_sq": fl.view(fl.v, i)}
                 for i in range(len(fl.params))} if self._adam_steps() > 0 else {}
        group = {"lr": self.lr, "betas": (self.beta1, self.beta2), "eps": self.eps, "weight_decay": self.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(fl.n_all))}
        return {"state": state, "param_groups": [group]}

    def _adam_steps(self):
        """the optimiser's own step counter (what torch stores per parameter as `step`)."""
        return self.step_count

    def _set_adam_steps(self, n):
        pass

    def load_optimizer_state_dict(self, sd):
        """accepts the reference's torch-optimizer layout (extra trailing parameters, e.g. the noise schedule's dummy
        parameter of losses.py:35-46, are ignored) or this trainer's compact {"m", "v", "step"}."""
        fl = self.flat
        if "state" in sd:
            steps = []
            for i in range(len(fl.params)):
                ent = sd["state"].get(fl.index[i])
                if ent is None:
                    fl.view(fl.m, i).zero_()
                    fl.view(fl.v, i).zero_()
                    continue
                fl.view(fl.m, i).copy_(ent["exp_avg"])
                fl.view(fl.v, i).copy_(ent["exp_avg_sq"])
                steps.append(int(ent["step"]))
            return max(steps) if steps else 0
        fl.m.copy_(sd["m"])
        fl.v.copy_(sd["v"])
        return int(sd["step"])

    def state_dict(self):
        return {"optimizer": self.optimizer_state_dict(), "model": self.model.state_dict(),
                "ema": self.ema.state_dict(), "step": self.step_count}

    def load_state_dict(self, sd):
        model_sd = sd["model"]
        if any(k.startswith("module.") for k in model_sd):  # DDP prefix (test/old_model/load_model_from_ckpt.py:57-62)
            model_sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in model_sd.items()}
        self.model.load_state_dict(model_sd)
        opt_step = self.load_optimizer_state_dict(sd["optimizer"])
        self.step_count = int(sd.get("step", opt_step))
        self._set_adam_steps(opt_step)
        for dst, src in zip(self.flat.ema_params(), sd["ema"]["shadow_params"]):
            dst.copy_(src)
        self.ema.num_updates = sd["ema"]["num_updates"]
        if self.flat.pc is not None:
            ops.cast(self.flat.p, torch.bfloat16, out=self.flat.pc)
        self.flat.refresh_compute_weights()


def save_checkpoint(ckpt_dir, state):
    """utils/utils.py:85-95: `state` is either the reference's dict of objects (optimizer / model / ema / step) or a
    SEDDTrainer; the file layout is the reference's either way."""
    if isinstance(state, SEDDTrainer):
        saved = state.state_dict()
    else:
        saved = {"optimizer": state["optimizer"].state_dict(), "model": state["model"].state_dict(),
                 "ema": state["ema"].state_dict(), "step": state["step"]}
    torch.save(saved, ckpt_dir)


def restore_checkpoint(ckpt_dir, state, device):
    """utils/utils.py:73-83 (returns `state` unchanged with a warning when the file is missing).  Files this package
    wrote hold tensors, numbers and dicts only and load with weights_only=True."""
    if not os.path.exists(ckpt_dir):
        os.makedirs(os.path.dirname(ckpt_dir) or ".", exist_ok=True)
        import logging
        logging.warning(f"No checkpoint found at {ckpt_dir}. Returned the same state as input")
        return state
    loaded = torch.load(ckpt_dir, map_location=device, weights_only=True)
    if isinstance(state, SEDDTrainer):
        state.load_state_dict(loaded)
        return state
    state["optimizer"].load_state_dict(loaded["optimizer"])
    state["ema"].load_state_dict(loaded["ema"])
    state["step"] = loaded["step"]
    return state


def shard_batch(batch, rank, world):
    """DistributedSampler-style contiguous shard of a global batch (data/process_full_plinder.py:396-398)."""
    per = batch.shape[0] // world
    return batch[rank * per:(rank + 1) * per]


def sharded_sampling(sampler, model, rank, world, gather=True, **kw):
    """Batch-sharded sampling: every rank samples its own shard with no communication during the steps; one final
    all_gather of the int64 indices."""
    x = sampler(model, **kw)
    if world == 1 or not gather:
        return x
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x)
    return torch.cat(out, 0)


class CrossAttTrainer(SEDDTrainer):
    """Training step of the dual-track cross-attention trainer (training/run_train_protlig_cross_att.py:192-333) on the
    flat-buffer / fused-optimiser machinery: `calc_loss(batch, task)` and `train_step(batch, task)` with the
    reference's quirks reproduced by default and switchable:

      * double_step=True  (Q2): the optimiser advances twice per iteration on the same clipped gradients
        (`scaler.step(optimizer)` in optimize_fn :247, then `optimizer.step()` :325); EMA is updated between the two;
      * per_sample_weighting=False (Q3): joint loss = mean(dsigma) * (mean(SE_prot) + mean(SE_lig)) / 2 (:289-294);
        True gives the per-sample weighting of processing/losses.py:29;
      * task freezing (:192-216): `protein_only` / `protein_given_ligand` freeze every parameter whose name contains
        'lig', `ligand_only` / `ligand_given_protein` those containing 'prot' — done before the flat buffers are built.

    batch: dict with int64 'prot_tokens' [B, Lp] and/or 'lig_tokens' [B, Ll].  The `*_given_*` tasks of the reference
    trainer pass raw sequence strings to the ESM / MolFormer encoders (network downloads) and are out of scope."""

    def __init__(self, model, noise, graph_prot, graph_lig, optim, task="joint", ema_decay=0.9999, sampling_eps=1e-3,
                 world_size=None, process_group=None, lr_schedule=None, double_step=True, per_sample_weighting=False,
                 hip_graph=False, grad_comm_dtype=None, zero1=None, wgrad_stream=True):
        # (wgrad_stream defaults to True here: the dual-track step is thousands of launches that each fill a fraction of
        # the chip at the reference's batch sizes — 36.7 -> 33.5 ms at batch 4, 71.8 -> 66.8 ms at batch 32, same bits)
        if task in ("protein_given_ligand", "protein_only"):
            self.freeze_model(model, "lig")
        elif task in ("ligand_given_protein", "ligand_only"):
            self.freeze_model(model, "prot")
        super().__init__(model, noise, graph_prot, optim, ema_decay=ema_decay, sampling_eps=sampling_eps,
                         world_size=world_size, process_group=process_group, overlap=False, lr_schedule=lr_schedule,
                         hip_graph=hip_graph, grad_comm_dtype=grad_comm_dtype, zero1=zero1, wgrad_stream=wgrad_stream)
        self._always_norm = True  # (this trainer reports the gradient norm whether or not it clips)
        self.graph_prot, self.graph_lig = graph_prot, graph_lig
        self.task, self.double_step, self.per_sample_weighting = task, double_step, per_sample_weighting
        self.sampling_eps = sampling_eps
        self.opt_steps = 0  # Adam's own step counter (advances twice per iteration with double_step)
        self._graphs_by_task, self._graph_calls_by_task = {}, {}

    def _adam_steps(self):
        return self.opt_steps

    def _set_adam_steps(self, n):
        """resume: Adam's bias correction continues from the optimiser's stored step (2x the iteration count under
        double_step), not from 0."""
        self.opt_steps = int(n)

    @staticmethod
    def freeze_model(model, freeze_what):
        for name, param in model.named_parameters():
            if freeze_what in name:
                param.requires_grad = False
        return model

    def calc_loss(self, batch, task, t=None, perturbed=None):
        """-> loss [B, 1] (single-track tasks) or (loss, loss_prot, loss_lig) for 'joint' (:250-297).
        perturbed: optional dict of pre-corrupted 'prot_tokens' / 'lig_tokens' (bit-exact replay of a noise draw)."""
        perturbed = perturbed or {}
        first = batch["prot_tokens"] if "prot_tokens" in batch else batch["lig_tokens"]
        dev = first.device
        if t is None:
            t = (1 - self.sampling_eps) * torch.rand(first.shape[0], device=dev) + self.sampling_eps
        sigma, dsigma = self.noise(t)
        sig = sigma.reshape(-1)

        def corrupt(key, graph):
            return perturbed[key] if key in perturbed else graph.sample_transition(batch[key], sigma[:, None])

        if task == "protein_only":
            pi = corrupt("prot_tokens", self.graph_prot)
            ls = self.model(protein_indices=pi, timesteps=sig, mode=task)
            return dsigma[:, None] * self.graph_prot.score_entropy(ls, sigma[:, None], pi, batch["prot_tokens"])
        if task == "ligand_only":
            li = corrupt("lig_tokens", self.graph_lig)
            ls = self.model(ligand_indices=li, timesteps=sig, mode=task)
            return dsigma[:, None] * self.graph_lig.score_entropy(ls, sigma[:, None], li, batch["lig_tokens"])
        if task == "joint":
            pi, li = corrupt("prot_tokens", self.graph_prot), corrupt("lig_tokens", self.graph_lig)
            lp, ll = self.model(protein_indices=pi, ligand_indices=li, timesteps=sig, mode=task)
            se_p = self.graph_prot.score_entropy(lp, sigma[:, None], pi, batch["prot_tokens"])
            se_l = self.graph_lig.score_entropy(ll, sigma[:, None], li, batch["lig_tokens"])
            if self.per_sample_weighting:
                lp_, ll_ = (dsigma[:, None] * se_p).mean(-1, keepdim=True), (dsigma[:, None] * se_l).mean(-1, keepdim=True)
                return (lp_ + ll_) / 2, lp_, ll_
            mp, ml = se_p.mean(), se_l.mean()
            return dsigma[:, None] * ((mp + ml) / 2), dsigma[:, None] * mp, dsigma[:, None] * ml
        raise NotImplementedError(f"Task {task} not implemented")

    def _kernel_steps(self):
        return 2 if self.double_step else 1

    def _host_advance(self, step=None):
        self.step_count += 1
        lr = self.current_lr(self.step_count - 1) if self.lr_schedule is not None else self.current_lr(self.step_count)
        self.ema.num_updates += 1
        decay = min(self.ema_decay, (1 + self.ema.num_updates) / (10 + self.ema.num_updates))
        first = self.opt_steps + 1
        self.opt_steps += 2 if self.double_step else 1  # Q2: the same clipped gradients twice; the EMA takes the first
        return lr, decay, first, self._kernel_steps()   # step's weights (one pass)

    def train_step(self, batch, task=None, t=None, perturbed=None):
        """:299-333.  Returns loss (or (loss, loss_prot, loss_lig) for 'joint') as 0-dim device tensors."""
        task = task or self.task
        fl = self.flat

        def body(t=None, **tensors):
            b = {k[2:]: v for k, v in tensors.items() if k.startswith("b_")}
            pert = {k[2:]: v for k, v in tensors.items() if k.startswith("p_")}
            fl.g.zero_()
            set_training(self.model, True)
            out = self.calc_loss(b, task, t=t, perturbed=pert)
            loss = out[0].mean() if task == "joint" else out.mean()
            self._backward(loss)
            if task == "joint":
                return loss, out[1].mean(), out[2].mean()
            return (loss,)

        inputs = {"b_" + k: v for k, v in batch.items() if torch.is_tensor(v)}
        inputs.update({"p_" + k: v for k, v in (perturbed or {}).items()})
        if t is not None:
            inputs["t"] = t
        if self.hip_graph:
            saved, self._graphs = self._graphs, self._graphs_by_task.setdefault(task, {})
            saved_calls, self._graph_calls = self._graph_calls, self._graph_calls_by_task.setdefault(task, {})
            try:
                out = self._graphed_step(body, inputs)
            finally:
                self._graphs, self._graph_calls = saved, saved_calls
        else:
            out = body(**inputs)
            if self.dp:
                self._exchange_gradients()
            self._launch_update(*self._host_advance())
            fl.refresh_compute_weights()
            out = tuple(o.detach() for o in out)
        return out if task == "joint" else out[0]
