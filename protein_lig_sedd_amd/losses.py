"""Drop-in for protlig_dd/processing/losses.py: get_loss_fn / get_optimizer / optimization_manager / get_step_fn.
The optimiser half (get_optimizer / optimization_manager / get_step_fn) keeps the reference's call shapes and state
dict but runs on the flat-buffer trainer of train.py: one fused clip -> Adam(W) -> EMA kernel per step.

loss_fn keeps the reference's signature and arithmetic (losses.py:11-30): t ~ U(eps, 1) -> (sigma, dsigma) ->
sample_transition -> model log-score -> score_entropy -> mean(dsigma * loss).  New optional keyword `u=` supplies the
corruption uniforms.  With `fused=True` (default) and a differentiable model output the score-entropy forward, its
gradient and the dsigma/M weighting run as ONE kernel (sedd_score_entropy_fwd with gw), the mean is a HIP reduction.
"""
import torch

from . import ops
from .model import utils as mutils


class _FusedLossFn(torch.autograd.Function):
    """loss = sum_r gw[b(r)] * SE[r] with gw = dsigma / M; gradient w.r.t. the log-score produced in the same pass."""

    @staticmethod
    def forward(ctx, log_score, graph, sigma, dsigma, x, x0):
        M = x.numel()
        gw = (dsigma / M).contiguous()
        se, ds = ops.score_entropy_fwd(graph, log_score, sigma, x, x0, gw=gw)
        ctx.save_for_backward(ds)
        return ops.batch_weighted_sum(se, gw)

    @staticmethod
    def backward(ctx, gout):
        (ds,) = ctx.saved_tensors
        # gout is the scalar upstream gradient (1 for loss.backward()): applied on the device — reading it on the host
        # would stall the launch queue at the start of every backward pass
        ds = ops.scale_(ds.clone(), gout.detach().reshape(1).float())  # (the saved tensor stays intact for a re-run)
        return ds, None, None, None, None, None


def get_loss_fn(noise, graph, train=True, sampling_eps=1e-3, lv=False, fused=True, conditioners=False):
    """conditioners=False is the reference as written: loss_fn accepts esm_cond / mol_cond and drops them (the call that
    would forward them is commented out, losses.py:26).  True forwards them to `SEDD.forward(indices, sigma, esm_cond,
    mol_cond)` (transformer.py:277) — what that signature and the DDP trainer's call (run_train_ddp_protlig_transf.py:284)
    are for."""

    def loss_fn(model, batch, cond=None, t=None, perturbed_batch=None, esm_cond=None, mol_cond=None, u=None):
        """Batch shape: [B, L] int64."""
        if t is None:
            if lv:
                raise NotImplementedError("Yeah I gotta do this later")
            t = (1 - sampling_eps) * torch.rand(batch.shape[0], device=batch.device) + sampling_eps
        sigma, dsigma = noise(t)
        if perturbed_batch is None:
            perturbed_batch = graph.sample_transition(batch, sigma[:, None], u=u)
        if conditioners and (esm_cond is not None or mol_cond is not None):
            model.train(train)
            log_score = model(perturbed_batch, sigma.reshape(-1), esm_cond, mol_cond)
        else:
            log_score_fn = mutils.get_score_fn(model, train=train, sampling=False)
            log_score = log_score_fn(perturbed_batch, sigma)
        if fused and log_score.requires_grad:
            return _FusedLossFn.apply(log_score, graph, sigma, dsigma, perturbed_batch, batch)
        loss = graph.score_entropy(log_score, sigma[:, None], perturbed_batch, batch)
        return (dsigma[:, None] * loss).mean()

    return loss_fn


class FusedAdam:
    """What `get_optimizer` returns: the hyper-parameters of torch.optim.Adam / AdamW (losses.py:35-46) in torch's
    `param_groups` shape, bound lazily to a `train.SEDDTrainer` — flat parameter / gradient / moment buffers and ONE
    fused clip -> Adam(W) -> EMA -> bf16-copy kernel (sedd_adamw_ema_step) instead of torch's per-parameter loops.
    `state_dict()` / `load_state_dict()` keep torch's checkpoint layout (utils/utils.py:73-95), also before binding (a
    restored state is parked and applied when the trainer is built on the first `step_fn` call)."""

    def __init__(self, params, name, lr, betas, eps, weight_decay):
        self.params = [p for p in params]
        self.name = name
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "params": self.params}]
        self.base_lr = lr
        self.trainer = None
        self._pending = None

    # ---- torch.optim.Optimizer surface the reference touches -------------------------------------------------------
    def state_dict(self):
        if self.trainer is None:
            return self._pending if self._pending is not None else {"state": {}, "param_groups": [
                {k: v for k, v in self.param_groups[0].items() if k != "params"} | {"params": list(range(len(self.params)))}]}
        return self.trainer.optimizer_state_dict()

    def load_state_dict(self, sd):
        if self.trainer is None:
            self._pending = sd
        else:
            self.trainer._set_adam_steps(self.trainer.load_optimizer_state_dict(sd))

    def zero_grad(self, set_to_none=True):
        if self.trainer is not None:
            self.trainer.flat.g.zero_()
        else:
            for p in self.params:
                p.grad = None

    def step(self):
        raise RuntimeError("FusedAdam is stepped by losses.get_step_fn's step_fn / optimize_fn (one fused kernel over "
                           "the flat buffers), not parameter by parameter")

    def bind(self, trainer):
        self.trainer = trainer
        if self._pending is not None:
            trainer._set_adam_steps(trainer.load_optimizer_state_dict(self._pending))
            self._pending = None


def get_optimizer(config, params):
    """losses.py:35-46: Adam / AdamW by `config.optim.optimizer`, NotImplementedError otherwise."""
    o = config.optim
    if o.optimizer not in ("Adam", "AdamW"):
        raise NotImplementedError(f'Optimizer {o.optimizer} not supported yet!')
    return FusedAdam(params, o.optimizer, float(o.lr), (o.beta1, o.beta2), o.eps, o.weight_decay)


def optimization_manager(config):
    """losses.py:49-71: warm-up -> clip -> optimiser step, here as one call into the bound trainer's fused step (the
    GradScaler is accepted and ignored: bf16 / fp32 need no loss scaling)."""

    def optimize_fn(optimizer, scaler, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                    grad_clip=config.optim.grad_clip):
        tr = getattr(optimizer, "trainer", None)
        if tr is None:
            raise RuntimeError("optimize_fn needs a FusedAdam bound to a trainer (losses.get_step_fn binds it on its first call)")
        tr.lr, tr.warmup, tr.grad_clip = float(lr), warmup, grad_clip
        tr.apply_gradients(step=step)

    optimize_fn.config = config
    return optimize_fn


def _bind_trainer(state, noise, graph, optimize_fn, accum=1):
    """Build the flat-buffer trainer for a reference-style `state` dict (run_train_ddp_protlig_transf.py:207-213:
    optimizer / scaler / model / noise / ema / step) once, and make the objects the caller already holds —
    state['optimizer'], state['ema'] — views onto it."""
    from .train import SEDDTrainer
    opt, model, ema = state['optimizer'], state['model'], state['ema']
    if not isinstance(opt, FusedAdam):
        raise TypeError("state['optimizer'] must come from protein_lig_sedd_amd.losses.get_optimizer")
    cfg = getattr(optimize_fn, "config", None)
    g = opt.param_groups[0]
    optim = dict(lr=opt.base_lr, beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                 optimizer=opt.name, warmup=cfg.optim.warmup if cfg is not None else 0,
                 grad_clip=cfg.optim.grad_clip if cfg is not None else -1.0)
    tr = SEDDTrainer(model, noise, graph, optim, ema_decay=ema.decay, ema=ema, accum=accum)
    tr.step_count = int(state['step'])
    tr._set_adam_steps(tr.step_count)
    opt.bind(tr)
    state['trainer'] = tr
    return tr


def get_step_fn(noise, graph, train, optimize_fn, accum, true_accumulation=False):
    """losses.py:74-117 on the fused path: the training branch is `SEDDTrainer.train_step` (loss -> backward into the flat
    gradient buffer -> RCCL all-reduce when data parallel -> one fused optimiser + EMA kernel), with the reference's
    return value and `state['step']` bookkeeping; the eval branch is the reference's EMA store / copy_to / restore
    around a no-grad loss.  `t=` / `perturbed_batch=` are optional replay arguments (parity tests)."""
    if accum != 1 and not true_accumulation:
        raise NotImplementedError("gradient accumulation: the reference's step_fn resets its counter on every call "
                                  "(losses.py:84), so it only ever worked for accum == 1; pass true_accumulation=True for "
                                  "the intended behaviour (k micro-steps of loss / k, one optimiser step)")
    loss_fn = get_loss_fn(noise, graph, train)

    def step_fn(state, batch, cond=None, esm_cond=None, mol_cond=None, t=None, perturbed_batch=None):
        tr = state.get('trainer') or _bind_trainer(state, noise, graph, optimize_fn, accum)
        model = state['model']
        if train:
            cfg = getattr(optimize_fn, "config", None)
            if cfg is not None:
                tr.lr, tr.warmup, tr.grad_clip = float(cfg.optim.lr), cfg.optim.warmup, cfg.optim.grad_clip
            tr.step_count = int(state['step'])
            loss = tr.train_step(batch, t=t, perturbed_batch=perturbed_batch)
            state['step'] = tr.step_count   # (advances on the accum-th micro-step only)
        else:
            with torch.no_grad():
                ema = state['ema']
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch, cond=cond, t=t, perturbed_batch=perturbed_batch).mean()
                ema.restore(model.parameters())
        return loss

    return step_fn
