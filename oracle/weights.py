"""Deterministic, order-independent weight recipe shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the product path imports this.

The reference zero-initialises every adaLN / final-linear weight (protlig_dd/model/transformer.py:141-142,
226-231), which makes attention/MLP numerically dead at init.  Parity fixtures therefore use random N(0, 0.02)
weights everywhere (LayerNorm scales 1 + N(0, 0.02)); each tensor is drawn from its own CPU generator seeded by
crc32(name), so the recipe does not depend on module registration order.
"""
import zlib

import torch


def seeded_tensor(name: str, shape, seed: int = 0, std: float = 0.02) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std
    leaf = name.split(".")
    # LayerNorm scale vectors: norm1.weight / norm2.weight / norm_final.weight / (cross_attn.)norm.weight
    if len(leaf) >= 2 and leaf[-1] == "weight" and leaf[-2].startswith("norm"):
        t = t + 1.0
    return t


def seeded_state_dict(shapes: dict, seed: int = 0, std: float = 0.02) -> dict:
    """shapes: {param_name: shape}. Returns {param_name: fp32 CPU tensor}."""
    out = {}
    for name in sorted(shapes):
        shape = shapes[name]
        if name.endswith("inv_freq") or name.endswith("num_batches_tracked"):
            continue
        out[name] = seeded_tensor(name, shape, seed, std)
    return out


def load_seeded(module: torch.nn.Module, seed: int = 0, std: float = 0.02) -> None:
    """Overwrite every parameter of `module` with the seeded recipe (buffers untouched)."""
    shapes = {n: tuple(p.shape) for n, p in module.named_parameters()}
    sd = seeded_state_dict(shapes, seed, std)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() == 0:
                continue  # noise modules' dummy scalar parameter
            p.copy_(sd[n].to(p.device, p.dtype))
