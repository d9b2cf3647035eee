"""Drop-in for protlig_dd/sampling/sampling.py: predictor registry, Euler / Analytic / None predictors, Denoiser,
get_pc_sampler / get_sampling_fn — with each update step running as ONE fused HIP kernel on the model's log-score
(exp -> reverse rate | staggered score x transposed transition -> Gumbel arg-max; csrc/graph_ops.hip).

Differences from the reference, all additive:
  * `u=` (optional) on every update_fn and `noise_seq=` on pc_sampler supply the uniform noise, so identical noise
    tensors give bit-identical indices (north_star); default None draws torch.rand on the device.
  * the score model may be either the dual-track cross model (keyword call with `mode=task`, sampling.py:64-76) or a
    plain SEDD (`model(x, sigma)`); the reference's loop only supports the former.
  * AnalyticPredictor accepts both its upstream positional signature `update_fn(score_fn, x, t, step_size)` and the
    keyword signature pc_sampler uses, so `predictor="analytic"` works through get_pc_sampler (in
    the reference that call raises TypeError).
"""
import abc
import inspect

import torch

from . import _lib as _L
from . import ops
from .catsample import sample_categorical

_PREDICTORS = {}


def register_predictor(cls=None, *, name=None):
    """A decorator for registering predictor classes (sampling.py:11-28)."""

    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _PREDICTORS:
            raise ValueError(f'Already registered model with name: {local_name}')
        _PREDICTORS[local_name] = cls
        return cls

    if cls is None:
        return _register
    return _register(cls)


def get_predictor(name):
    return _PREDICTORS[name]


def _is_dual_track(model):
    try:
        params = inspect.signature(model.forward).parameters
    except (TypeError, ValueError):
        return False
    return "mode" in params and "timesteps" in params


def model_log_score(score_model, x, sigma, task=None, prot_indices=None, lig_indices=None):
    """The model call of sampling.py:64-76 / :110-122, returning the LOG score."""
    if hasattr(score_model, "eval"):
        score_model.eval()
    if _is_dual_track(score_model):
        if task in ("ligand_given_protein", "ligand_only"):
            return score_model(ligand_indices=x, protein_indices=prot_indices, timesteps=sigma.reshape(-1), mode=task)
        elif task in ("protein_given_ligand", "protein_only"):
            return score_model(protein_indices=x, ligand_indices=lig_indices, timesteps=sigma.reshape(-1), mode=task)
        raise ValueError(f"Invalid mode: {task}")
    return score_model(x, sigma.reshape(-1))


class Predictor(abc.ABC):
    """The abstract class for a predictor algorithm (sampling.py:36-56)."""

    def __init__(self, graph, noise):
        super().__init__()
        self.graph = graph
        self.noise = noise

    @abc.abstractmethod
    def update_fn(self, score_model, x, t, step_size, task=None, prot_indices=None, lig_indices=None, u=None):
        pass


@register_predictor(name="euler")
class EulerPredictor(Predictor):
    """sampling.py:59-80"""

    def update_fn(self, score_model, x, t, step_size, task=None, prot_indices=None, lig_indices=None, u=None):
        sigma, dsigma = self.noise(t)
        log_score = model_log_score(score_model, x, sigma, task, prot_indices, lig_indices)
        return self.graph.predictor_step(_L.STEP_EULER, log_score, x, step_size * dsigma, u=u)


@register_predictor(name="none")
class NonePredictor(Predictor):
    """sampling.py:82-85"""

    def update_fn(self, score_fn=None, x=None, t=None, step_size=None, *args, score_model=None, **kw):
        return x


@register_predictor(name="analytic")
class AnalyticPredictor(Predictor):
    """sampling.py:88-99"""

    def update_fn(self, score_fn=None, x=None, t=None, step_size=None, task=None, prot_indices=None, lig_indices=None,
                  u=None, score_model=None):
        curr_sigma = self.noise(t)[0]
        next_sigma = self.noise(t - step_size)[0]
        dsigma = curr_sigma - next_sigma
        if score_model is not None:  # pc_sampler's keyword call: fused kernel on the model's log-score
            log_score = model_log_score(score_model, x, curr_sigma, task, prot_indices, lig_indices)
            return self.graph.predictor_step(_L.STEP_ANALYTIC, log_score, x, dsigma, u=u)
        # upstream call: score_fn returns the (exponentiated) score -> Graph-method kernels
        score = score_fn(x, curr_sigma)
        stag_score = self.graph.staggered_score(score, dsigma)
        probs = stag_score * self.graph.transp_transition(x, dsigma)
        return sample_categorical(probs, u=u)


class Denoiser:
    """sampling.py:102-132"""

    def __init__(self, graph, noise):
        self.graph = graph
        self.noise = noise

    def update_fn(self, score_model, x, t, step_size=None, task=None, prot_indices=None, lig_indices=None, u=None):
        sigma = self.noise(t)[0]
        log_score = model_log_score(score_model, x, sigma, task, prot_indices, lig_indices)
        return self.graph.predictor_step(_L.STEP_DENOISE, log_score, x, sigma, u=u)


def get_sampling_fn(config, graph, noise, batch_dims, eps, device):
    """sampling.py:135-146"""
    return get_pc_sampler(graph=graph, noise=noise, batch_dims=batch_dims, predictor=config.sampling.predictor,
                          steps=config.sampling.steps, denoise=config.sampling.noise_removal, eps=eps, device=device)


class _GraphedStep:
    """One predictor update captured as a HIP graph: `x <- update(proj(x), t)` over static buffers, replayed once per
    sampler step.  A small model's step is a few hundred launches of a few microseconds each and the Python between them
    costs more than the kernels; the replay is one launch.  Everything that changes from step to step lives in device
    memory (x, t, and the explicit noise when the caller passes one); the step size is the same Python float every step."""

    def __init__(self, fn, x, t, u):
        self.x, self.t = x.clone(), t.clone()
        self.u = None if u is None else u.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):  # (lazy scratch allocations and the allocator's pools settle outside the capture)
            for _ in range(2):
                fn(self.x, self.t, self.u)
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self.x.copy_(fn(self.x, self.t, self.u))
        except RuntimeError as e:  # (torch.AcceleratorError is a RuntimeError)
            self.graph = None      # drop the half-built graph object and the static buffers with it
            self.x = self.t = self.u = None
            raise RuntimeError("hip_graph=True: the predictor update could not be captured — proj_fun or the model call "
                               "does host-side work under capture (a Python list / bool-mask index, .item(), a "
                               f"host-to-device copy ...): {e}") from e

    def run(self, x, t_seq, noise_seq, steps):
        self.x.copy_(x)
        for i in range(steps):
            self.t.copy_(t_seq[i])
            if self.u is not None:
                self.u.copy_(noise_seq[i])
            self.graph.replay()
        return self.x.clone()


def get_pc_sampler(graph, noise, batch_dims, predictor, steps, denoise=True, eps=1e-5, device=torch.device('cpu'),
                   proj_fun=lambda x: x, hip_graph=False):
    """sampling.py:149-175.  hip_graph=True (not in the reference): the predictor update is captured once per (model,
    task, conditioning, noise form) as a HIP graph and replayed `steps` times — same kernels, same tokens for the same
    noise; for launch-bound (small) models.  It needs proj_fun and the model's forward to be capturable (device-side
    work only) and takes noise_seq as explicit tensors or None (torch's generator), not as a PhiloxStream (whose offsets
    are host integers)."""
    predictor = get_predictor(predictor)(graph, noise)
    projector = proj_fun
    denoiser = Denoiser(graph, noise)
    graphed = {}

    @torch.no_grad()
    def pc_sampler(model, task=None, prot_indices=None, lig_indices=None, noise_seq=None, x_init=None):
        """noise_seq: optional sequence of `steps` (+1 when denoising) uniform tensors, one per update — or
        ops.PhiloxStream(seed), with which every update generates its noise inside the predictor kernel."""
        x = graph.sample_limit(*batch_dims).to(device) if x_init is None else x_init.to(device)
        timesteps = torch.linspace(1, eps, steps + 1, device=device)
        dt = (1 - eps) / steps
        ones = torch.ones(x.shape[0], 1, device=device)
        mark = getattr(model, "sampling_step", None)  # dual-track model: per-step cache of the conditioning context
        mark = mark if callable(mark) else (lambda i: None)
        use_graph = hip_graph and x.is_cuda and steps > 0
        if use_graph and isinstance(noise_seq, ops.PhiloxStream):
            raise ValueError("hip_graph=True takes noise_seq as tensors or None, not a PhiloxStream")
        try:
            if use_graph:
                def one(xs, ts, us):
                    return predictor.update_fn(score_model=model, x=projector(xs), t=ts, step_size=dt, task=task,
                                               prot_indices=prot_indices, lig_indices=lig_indices, u=us)

                u0 = None if noise_seq is None else noise_seq[0]
                # (parameter versions: a capture holds the compute-dtype weight copies that were cached when it was made;
                # weights replaced through autograd-visible writes — load_state_dict, ema.copy_to — bump a version and
                # force a new capture, the trainer's in-place flat-buffer updates keep the cached views current)
                wsig = sum(p._version for p in model.parameters()) if hasattr(model, "parameters") else 0

                def tsig(ten):  # address AND extent: the allocator may hand a recycled address to a tensor of another shape
                    return None if ten is None else (ten.data_ptr(), tuple(ten.shape), ten.dtype, ten._version)

                key = (task, tsig(prot_indices), tsig(lig_indices), tuple(x.shape),
                       None if u0 is None else tuple(u0.shape), wsig)
                ent = graphed.get(key)
                # the model is compared by identity through a weak reference (an id() can be recycled by a new object)
                if ent is None or ent[0]() is not model:
                    mark(None)  # (the per-step context cache is host state: the captured step recomputes the context)
                    graphed.clear()
                    import weakref
                    ent = graphed[key] = (weakref.ref(model), _GraphedStep(one, x, timesteps[0] * ones, u0))
                x = ent[1].run(x, timesteps[:steps, None, None] * ones, noise_seq, steps)
            for i in range(0 if use_graph else steps):
                mark(i)
                t = timesteps[i] * ones
                x = projector(x)
                u = None if noise_seq is None else noise_seq[i]
                x = predictor.update_fn(score_model=model, x=x, t=t, step_size=dt, task=task, prot_indices=prot_indices,
                                        lig_indices=lig_indices, u=u)
            if denoise:
                mark(steps)
                x = projector(x)
                t = timesteps[-1] * ones
                u = None if noise_seq is None else noise_seq[steps]
                x = denoiser.update_fn(score_model=model, x=x, t=t, step_size=dt, task=task, prot_indices=prot_indices,
                                       lig_indices=lig_indices, u=u)
        finally:
            mark(None)  # the context cache never outlives the run, also when an update raises
        return x

    return pc_sampler
