"""predictor_step at the cfg-5 shard (512 x 1024 rows, V = 26, bf16 log-scores), by mask fraction: us and GB/s of
algorithmic bytes M (V e + 4 V + 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops, graph_lib, _lib as L
B, Ln, V = 512, 1024, 26
M = B * Ln
g = graph_lib.Absorbing(V - 1)
logits = torch.randn(B, Ln, V, device="cuda").bfloat16()
u = torch.rand(B, Ln, V, device="cuda")
sig = torch.rand(B, device="cuda") + 0.1
print({k: v for k, v in os.environ.items() if k.startswith("SEDD_")})
for frac in (0.5, 1.0, 0.1):
    x = torch.randint(0, V - 1, (B, Ln), device="cuda"); x[torch.rand(B, Ln, device="cuda") < frac] = V - 1
    for nm, stp, ph in (("analytic", L.STEP_ANALYTIC, 0), ("euler", L.STEP_EULER, 0), ("analytic+philox", L.STEP_ANALYTIC, 1),
                        ("rand+analytic", L.STEP_ANALYTIC, 2)):
        if ph == 1: fn = lambda: ops.predictor_step_philox(g, stp, logits, x, sig, ops.PhiloxNoise(1, 3))
        elif ph == 2: fn = lambda: ops.predictor_step(g, stp, logits, x, sig, torch.rand(B, Ln, V, device="cuda"))
        else: fn = lambda: ops.predictor_step(g, stp, logits, x, sig, u)
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        nb = M * (V * 2 + (0 if ph == 1 else V * 4) + 16)
        print(f"masked {frac:.1f} {nm:16s} {us:6.1f} us  {nb/us/1e3:6.0f} GB/s  {nb/us/1e3/80:.1f}% of 8 TB/s", flush=True)
