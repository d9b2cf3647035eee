"""Drop-in for protlig_dd/processing/losses.py: get_loss_fn / get_optimizer / optimization_manager / get_step_fn.

loss_fn keeps the reference's signature and arithmetic (losses.py:11-30): t ~ U(eps, 1) -> (sigma, dsigma) ->
sample_transition -> model log-score -> score_entropy -> mean(dsigma * loss).  New optional keyword `u=` supplies the
corruption uniforms.  With `fused=True` (default) and a differentiable model output the score-entropy forward, its
gradient and the dsigma/M weighting run as ONE kernel (sedd_score_entropy_fwd with gw), the mean is a HIP reduction.
"""
import numpy as np
import torch
import torch.optim as optim

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


def get_loss_fn(noise, graph, train=True, sampling_eps=1e-3, lv=False, fused=True):
    def loss_fn(model, batch, cond=None, t=None, perturbed_batch=None, esm_cond=None, mol_cond=None, u=None):
        """Batch shape: [B, L] int64."""
        if t is None:
            if lv:
                raise NotImplementedError("Yeah I gotta do this later")
            t = (1 - sampling_eps) * torch.rand(batch.shape[0], device=batch.device) + sampling_eps
        sigma, dsigma = noise(t)
        if perturbed_batch is None:
            perturbed_batch = graph.sample_transition(batch, sigma[:, None], u=u)
        log_score_fn = mutils.get_score_fn(model, train=train, sampling=False)
        log_score = log_score_fn(perturbed_batch, sigma)
        if fused and log_score.requires_grad:
            return _FusedLossFn.apply(log_score, graph, sigma, dsigma, perturbed_batch, batch)
        loss = graph.score_entropy(log_score, sigma[:, None], perturbed_batch, batch)
        return (dsigma[:, None] * loss).mean()

    return loss_fn


def get_optimizer(config, params):
    """losses.py:35-46"""
    if config.optim.optimizer == 'Adam':
        return optim.Adam(params, lr=config.optim.lr, betas=(config.optim.beta1, config.optim.beta2),
                          eps=config.optim.eps, weight_decay=config.optim.weight_decay)
    elif config.optim.optimizer == 'AdamW':
        return optim.AdamW(params, lr=config.optim.lr, betas=(config.optim.beta1, config.optim.beta2),
                           eps=config.optim.eps, weight_decay=config.optim.weight_decay)
    raise NotImplementedError(f'Optimizer {config.optim.optimizer} not supported yet!')


def optimization_manager(config):
    """losses.py:49-71 (torch optimizer path; the fused HIP AdamW+EMA step lives in train.py)."""

    def optimize_fn(optimizer, scaler, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                    grad_clip=config.optim.grad_clip):
        scaler.unscale_(optimizer)
        if warmup > 0:
            for g in optimizer.param_groups:
                g['lr'] = lr * np.minimum(step / warmup, 1.0)
        if grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        scaler.step(optimizer)
        scaler.update()

    return optimize_fn


def get_step_fn(noise, graph, train, optimize_fn, accum):
    """losses.py:74-117"""
    loss_fn = get_loss_fn(noise, graph, train)

    def step_fn(state, batch, cond=None, esm_cond=None, mol_cond=None):
        accum_iter = 0
        total_loss = 0
        model = state['model']
        if train:
            optimizer = state['optimizer']
            scaler = state['scaler']
            loss = loss_fn(model, batch, cond=cond, esm_cond=esm_cond, mol_cond=mol_cond).mean() / accum
            scaler.scale(loss).backward()
            accum_iter += 1
            total_loss += loss.detach()
            if accum_iter == accum:
                accum_iter = 0
                state['step'] += 1
                optimize_fn(optimizer, scaler, model.parameters(), step=state['step'])
                state['ema'].update(model.parameters())
                optimizer.zero_grad()
                loss = total_loss
                total_loss = 0
        else:
            with torch.no_grad():
                ema = state['ema']
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch, cond=cond).mean()
                ema.restore(model.parameters())
        return loss

    return step_fn
