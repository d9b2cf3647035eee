"""K sweep of sedd_gemm_nt at a fixed output size: separates the per-K-step cost from the per-tile fixed cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for (M, N) in ((32768, 2304), (32768, 768), (8192, 2048), (4096, 4096)):
    for od in (torch.bfloat16, torch.float32):
        for K in (128, 256, 768, 1536, 3072):
            A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
            C = torch.zeros(M, N, device="cuda", dtype=od)
            ms = timeit(lambda: ops.gemm_nt(A, B, out=C))
            print(f"M={M} N={N} K={K} out={'bf16' if od == torch.bfloat16 else 'fp32'}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
