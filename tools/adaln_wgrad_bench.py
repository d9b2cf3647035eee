import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = "cuda"
A = torch.randn(64, 4608, device=dev).bfloat16(); B = torch.randn(64, 128, device=dev).bfloat16()
C = torch.zeros(4608, 128, device=dev)
ref = (A.double().T @ B.double())
ops.gemm_tn(A, B, out=C, accumulate=True); ops.gemm_tn(A, B, out=C, accumulate=True)
torch.cuda.synchronize()
print("rel err", float((C.double() - 2 * ref).norm() / (2 * ref).norm()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.gemm_tn(A, B, out=C, accumulate=True)
e1.record(); torch.cuda.synchronize()
print("adaLN wgrad TN [4608 x 128, K=64]:", e0.elapsed_time(e1) / 50 * 1e3, "us")
