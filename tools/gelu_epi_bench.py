"""fc1-forward shape (M=32768, N=3072, K=768): cost of the epilogue variants, back-to-back launches (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = torch.device("cuda", 0)
M, N, K = 32768, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16(); B = (0.05 * torch.randn(N, K, device=dev)).bfloat16()
bias = torch.randn(N, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * M * N * K
for name, fn in (("plain", lambda: ops.gemm_nt(A, B)), ("bias", lambda: ops.gemm_nt(A, B, bias=bias)),
                 ("bias+gelu", lambda: ops.gemm_nt(A, B, bias=bias, gelu=True)),
                 ("bias+gelu+pre", lambda: ops.gemm_nt(A, B, bias=bias, gelu=True, want_pre=True))):
    us = t(fn)
    print(f"{name:16s} {us:7.1f} us  {fl/us/1e6:6.0f} TF/s", flush=True)
