"""Drop-in for protlig_dd/model/ema.py with the update on the HIP EMA kernel (one launch per parameter, or ONE launch
when the parameters are views of a single flat buffer as the trainer in train.py lays them out)."""
import torch

from .. import ops


def _drop_weight_cache(param):
    """forget the cached compute-dtype copy of a parameter whose values were just overwritten in place."""
    if getattr(param, "_sedd_wc", None) is not None:
        param._sedd_wc = None


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []
        self._before_read = None  # set by a sharded trainer: gathers the shadow's shards before it is read as a whole

    def current_decay(self):
        """decay that the NEXT update will use after incrementing num_updates (ema.py:38-42)."""
        decay = self.decay
        if self.num_updates is not None:
            decay = min(decay, (1 + self.num_updates + 1) / (10 + self.num_updates + 1))
        return decay

    def update(self, parameters):
        """ema.py:32-52"""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        one_minus_decay = 1.0 - decay
        with torch.no_grad():
            parameters = [p for p in parameters if p.requires_grad]
            for s_param, param in zip(self.shadow_params, parameters):
                if s_param.is_cuda:
                    ops.ema_update_(s_param, param.detach().contiguous(), one_minus_decay)
                else:
                    raise RuntimeError("ExponentialMovingAverage.update needs CUDA(HIP) parameters: no CPU fallback")

    def copy_to(self, parameters):
        """ema.py:54-66.  The copy goes through `param.copy_` (not `param.data.copy_`): it bumps the parameter's
        version counter, which is what the compute-dtype weight cache of model/layers.py is keyed on — a forward after
        copy_to must see the EMA weights in every Linear, not a stale bf16 copy of the training weights."""
        parameters = [p for p in parameters if p.requires_grad]
        if self._before_read is not None:
            self._before_read()
        with torch.no_grad():
            for s_param, param in zip(self.shadow_params, parameters):
                param.copy_(s_param)
                _drop_weight_cache(param)

    def store(self, parameters):
        self.collected_params = [param.detach().clone() for param in parameters]

    def restore(self, parameters):
        """ema.py:76-91 (same version-bumping copy as copy_to)."""
        with torch.no_grad():
            for c_param, param in zip(self.collected_params, parameters):
                param.copy_(c_param)
                _drop_weight_cache(param)

    def state_dict(self):
        if self._before_read is not None:
            self._before_read()
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        new = state_dict['shadow_params']
        # same values as the reference's plain assignment (ema.py:96-99), but written INTO the existing shadow tensors
        # when they line up: under the flat-buffer trainer they are views of one buffer that the fused kernel updates
        if len(new) == len(self.shadow_params) and all(a.shape == b.shape for a, b in zip(new, self.shadow_params)):
            with torch.no_grad():
                for dst, src in zip(self.shadow_params, new):
                    if dst is not src:
                        dst.copy_(src)
        else:
            self.shadow_params = new
