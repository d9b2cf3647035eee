"""fused Linear->GELU data gradient: the dGELU epilogue (pre-activation in) against the multiply epilogue (gelu' in),
and the two forward epilogues that feed them, at the cfg-2 MLP shape (M 32768, hidden 3072, D 768)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_lig_sedd_amd import ops
M, Hd, D = 32768, 3072, 768
torch.manual_seed(0)
dev = "cuda"
X = (0.5 * torch.randn(M, D, device=dev)).bfloat16()
W1 = (0.05 * torch.randn(Hd, D, device=dev)).bfloat16()
b1 = torch.randn(Hd, device=dev) * 0.1
dY = (0.3 * torch.randn(M, D, device=dev)).bfloat16()
W2 = (0.05 * torch.randn(D, Hd, device=dev)).bfloat16()
g, u = ops.gemm_nt(X, W1, bias=b1, gelu=True, want_pre=True)
_, gp = ops.gemm_nt(X, W1, bias=b1, gelu=True, want_pre="grad")
cs = torch.zeros(Hd, device=dev)
cases = (("fwd gelu + pre", lambda: ops.gemm_nt(X, W1, bias=b1, gelu=True, want_pre=True)),
         ("fwd gelu + gelu'", lambda: ops.gemm_nt(X, W1, bias=b1, gelu=True, want_pre="grad")),
         ("fwd gelu only", lambda: ops.gemm_nt(X, W1, bias=b1, gelu=True)),
         ("fwd bias only", lambda: ops.gemm_nt(X, W1, bias=b1)),
         ("bwd dgelu (nn)", lambda: ops.gemm_nn_dgelu(dY, W2, u, colsum=cs)),
         ("bwd dmul (nn)", lambda: ops.gemm_dmul(dY, W2, gp, colsum=cs)),
         ("bwd plain nn", lambda: ops.gemm_nn(dY, W2)))
for rep in range(2):
    for name, fn in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        print(f"{name:20s} {us:7.1f} us  {2.0 * M * Hd * D / us / 1e6:6.0f} TF/s", flush=True)
