"""Drop-in for protlig_dd/model/utils.py (get_model_fn / get_score_fn)."""
import inspect

import torch


def _accepts_task(model):
    try:
        params = inspect.signature(model.forward).parameters
    except (TypeError, ValueError):
        return False
    return "task" in params or any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values())


def get_model_fn(model, train=False):
    """model/utils.py:5-37.  The reference forwards prot_seq / lig_seq / task keyword arguments that its own
    SEDD.forward does not accept; they are passed only to models whose forward takes them."""
    takes_task = _accepts_task(model)

    def model_fn(x, sigma, prot_seq=None, lig_seq=None, task="ligand_given_protein"):
        if train:
            model.train()
        else:
            model.eval()
        if takes_task:
            return model(x, sigma, prot_seq=prot_seq, lig_seq=lig_seq, task=task)
        return model(x, sigma)

    return model_fn


def get_score_fn(model, train=False, sampling=False):
    """model/utils.py:40-56 (conditioning kwargs are dropped exactly as in the reference, quirk Q4)."""
    if sampling:
        assert not train, "Must sample in eval mode"
    model_fn = get_model_fn(model, train=train)

    def score_fn(x, sigma, prot_seq=None, lig_seq=None, task="ligand_given_protein"):
        sigma = sigma.reshape(-1)
        score = model_fn(x, sigma, prot_seq=None, lig_seq=None, task="ligand_given_protein")
        if sampling:
            return score.float().exp()  # true score (model/utils.py:50-52); tiny [B,L,V] op kept for API parity only
        return score

    return score_fn
