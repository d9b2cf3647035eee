"""the two HBM kernels VERDICT r2 names, alone, at the cfg-5 shard (for rocprofv3 --pmc / --kernel-trace passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops, graph_lib, _lib as L
dev, V, B, Ln = "cuda", 26, 512, 1024
g = graph_lib.Absorbing(V - 1)
x = torch.randint(0, V, (B, Ln), device=dev); x[torch.rand(B, Ln, device=dev) < 0.5] = V - 1
x0 = torch.randint(0, V - 1, (B, Ln), device=dev)
logits = torch.randn(B, Ln, V, device=dev).bfloat16()
u = torch.rand(B, Ln, V, device=dev)
sig = torch.rand(B, device=dev) + 0.1
for _ in range(20):
    ops.score_entropy_fwd(g, logits, sig, x, x0, gw=sig)
    ops.predictor_step(g, L.STEP_ANALYTIC, logits, x, sig, u)
torch.cuda.synchronize()
