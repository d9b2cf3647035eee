"""predictor_step only (for PMC counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops, graph_lib, _lib as L
B, Ln, V = 512, 1024, 26
g = graph_lib.Absorbing(V - 1)
x = torch.randint(0, V, (B, Ln), device="cuda"); x[torch.rand(B, Ln, device="cuda") < 0.5] = V - 1
logits = torch.randn(B, Ln, V, device="cuda").bfloat16()
u = torch.rand(B, Ln, V, device="cuda")
sig = torch.rand(B, device="cuda") + 0.1
for _ in range(5):
    ops.predictor_step(g, L.STEP_ANALYTIC, logits, x, sig, u)
torch.cuda.synchronize()
