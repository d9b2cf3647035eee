"""Drop-in for protlig_dd/utils/catsample.py: Gumbel-max categorical sampling on the HIP kernel.

`u=` (new, optional) supplies the uniform noise so that identical noise gives bit-identical indices."""
import torch

from . import ops


def sample_categorical(categorical_probs, method="hard", u=None):
    """catsample.py:10-15"""
    if method == "hard":
        if u is None:
            u = torch.rand(categorical_probs.shape, dtype=torch.float32, device=categorical_probs.device)
        return ops.gumbel_argmax(categorical_probs, u)
    raise ValueError(f"Method {method} for sampling categorical variables is not valid.")
