"""Drop-in for protlig_dd/processing/graph_lib.py: the token-graph algebra of SEDD with every tensor operation on
the HIP kernels of csrc/graph_ops.hip (C ABI: include/sedd_hip.h).

Same classes, method names, argument meaning and errors as the reference.  New optional keyword arguments (default
None = draw with torch's RNG on the device): `u=` / `r=` noise tensors, so identical noise gives bit-identical
indices.
"""
import abc

import torch

from . import _lib as _L
from . import ops
from .catsample import sample_categorical


def get_graph(config, device=None, tokens=None):
    """graph_lib.py:11-19"""
    if tokens is None:
        tokens = config.tokens
    if config.graph.type == "uniform":
        return Uniform(tokens)
    elif config.graph.type == "absorb":
        return Absorbing(tokens)
    raise ValueError(f"Graph {config.graph.type} not valid")


class _ScoreEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, score, sigma, x, x0):
        out, _ = ops.score_entropy_fwd(graph, score, sigma, x, x0)
        ctx.graph = graph
        ctx.save_for_backward(score, sigma, x, x0)
        return out

    @staticmethod
    def backward(ctx, gout):
        score, sigma, x, x0 = ctx.saved_tensors
        return None, ops.score_entropy_bwd(ctx.graph, score, sigma, x, x0, gout), None, None, None


class Graph(abc.ABC):
    """graph_lib.py:29-121"""
    code = -1

    @property
    def dim(self):
        pass

    @property
    def absorb(self):
        pass

    @abc.abstractmethod
    def rate(self, i):
        pass

    def transp_rate(self, i):
        return ops.transp_rate(self, i)

    @abc.abstractmethod
    def transition(self, i, sigma):
        pass

    def transp_transition(self, i, sigma):
        return ops.transp_transition(self, i, sigma)

    def reverse_rate(self, i, score):
        """graph_lib.py:79-93"""
        return ops.reverse_rate(self, i, score)

    def sample_rate(self, i, rate, u=None):
        """graph_lib.py:95-96: sample_categorical(onehot(i) + rate)"""
        probs = torch.nn.functional.one_hot(i, num_classes=self.dim).to(rate) + rate
        return sample_categorical(probs, u=u)

    def staggered_score(self, score, dsigma):
        return ops.staggered_score(self, score, dsigma)

    @abc.abstractmethod
    def sample_limit(self, *batch_dims):
        pass

    def score_entropy(self, score, sigma, x, x0):
        """graph_lib.py:170-197 / :274-314; differentiable w.r.t. score."""
        if score.requires_grad:
            return _ScoreEntropyFn.apply(self, score, sigma, x, x0)
        return ops.score_entropy_fwd(self, score, sigma, x, x0)[0]

    def predictor_step(self, step, logscore, x, coef, u=None):
        """Fused Euler / Analytic / Denoiser update (sampling.py:61-132) on the model's log-score."""
        V = logscore.shape[-1]
        Vu = V - 1 if (self.absorb and step == _L.STEP_DENOISE) else V
        if isinstance(u, ops.PhiloxNoise):  # opt-in: noise generated inside the kernel
            return ops.predictor_step_philox(self, step, logscore, x, coef, u)
        if u is None:
            u = torch.rand(*x.shape, Vu, dtype=torch.float32, device=x.device)
        return ops.predictor_step(self, step, logscore, x, coef, u)


class Uniform(Graph):
    """graph_lib.py:124-197"""
    code = _L.UNIFORM

    def __init__(self, dim):
        self._dim = dim

    @property
    def dim(self):
        return self._dim

    @property
    def absorb(self):
        return False

    def rate(self, i):
        return ops.transp_rate(self, i)  # symmetric graph: rate == transp_rate (graph_lib.py:144-145)

    def transition(self, i, sigma):
        return ops.transp_transition(self, i, sigma)

    def sample_transition(self, i, sigma, u=None, r=None):
        """graph_lib.py:156-160"""
        if u is None:
            u = torch.rand(i.shape, dtype=torch.float32, device=i.device)
        if r is None:
            r = torch.randint_like(i, self.dim)
        return ops.sample_transition(self, i, sigma, u, r)

    def sample_limit(self, *batch_dims):
        return torch.randint(0, self.dim, batch_dims)


class Absorbing(Graph):
    """graph_lib.py:200-314"""
    code = _L.ABSORB

    def __init__(self, dim):
        super().__init__()
        self._dim = dim

    @property
    def dim(self):
        return self._dim + 1

    @property
    def absorb(self):
        return True

    def rate(self, i):
        """graph_lib.py:213-216 (int64 like the reference)"""
        F = torch.nn.functional
        return F.one_hot((self.dim - 1) * torch.ones_like(i), num_classes=self.dim) - F.one_hot(i, num_classes=self.dim)

    def transp_rate(self, i):
        return ops.transp_rate(self, i).long()  # graph_lib.py:218-221 returns int64

    def transition(self, i, sigma):
        pass

    @staticmethod
    def curriculum_factor(training_step, preschool=100, elementary=500, middle=1000, high=1500, adult=2000):
        """graph_lib.py:236-249"""
        if training_step < preschool:
            return 3
        elif training_step < elementary:
            return 3 - min(1.0, training_step / elementary)
        elif training_step < middle:
            return 2
        elif training_step < high:
            return 2 - min(1.0, training_step / high) * 0.8
        elif training_step < adult:
            return 1.2
        return 1

    def sample_transition_curriculum(self, i, sigma, training_step, preschool=100, elementary=500, middle=1000,
                                     high=1500, adult=2000, u=None):
        """graph_lib.py:236-256"""
        f = self.curriculum_factor(training_step, preschool, elementary, middle, high, adult)
        return self.sample_transition(i, sigma * f, u=u)

    def sample_transition(self, i, sigma, u=None):
        """graph_lib.py:258-262"""
        if u is None:
            u = torch.rand(i.shape, dtype=torch.float32, device=i.device)
        return ops.sample_transition(self, i, sigma, u)

    def sample_limit(self, *batch_dims):
        return (self.dim - 1) * torch.ones(*batch_dims, dtype=torch.int64)
