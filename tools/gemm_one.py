"""one GEMM shape, a few launches (for PMC counter passes): python tools/gemm_one.py M N K [tn]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
tn = len(sys.argv) > 4
if tn:
    A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
    C = torch.zeros(M, N, device="cuda")
    for _ in range(5): ops.gemm_tn(A, B, out=C, accumulate=True)
else:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): ops.gemm_nt(A, B, out=C)
torch.cuda.synchronize()
