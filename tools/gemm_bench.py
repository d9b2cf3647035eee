"""Micro-benchmark of sedd_gemm_nt on the shapes of the cfg-2 train step (random bf16 data, HIP events)."""
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

M = 32768
shapes = [("qkv fwd", M, 2304, 768, torch.bfloat16, False), ("out fwd", M, 768, 768, torch.bfloat16, False),
          ("fc1 fwd", M, 3072, 768, torch.bfloat16, False), ("fc2 fwd", M, 768, 3072, torch.bfloat16, False),
          ("fc1 wgrad", 3072, 768, M, torch.float32, True), ("fc2 wgrad", 768, 3072, M, torch.float32, True),
          ("qkv wgrad", 2304, 768, M, torch.float32, True), ("4096^3", 4096, 4096, 4096, torch.bfloat16, False)]
for name, m, n, k, od, acc in shapes:
    A = torch.randn(m, k, device="cuda").bfloat16(); B = torch.randn(n, k, device="cuda").bfloat16()
    C = torch.zeros(m, n, device="cuda", dtype=od)
    ms = timeit(lambda: ops.gemm_nt(A, B, out=C, accumulate=acc))
    print(f"{name:10s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
x = torch.randn(M, 3072, device="cuda").bfloat16()
ms = timeit(lambda: ops.transpose(x)); print(f"transpose [{M},3072] bf16: {ms*1e3:.1f} us  {2*x.numel()*2/ms/1e6:.0f} GB/s")
for name, m, n, k in (("fc1 wgrad TN", 3072, 768, M), ("fc2 wgrad TN", 768, 3072, M), ("qkv wgrad TN", 2304, 768, M), ("out wgrad TN", 768, 768, M)):
    A = torch.randn(k, m, device="cuda").bfloat16(); B = torch.randn(k, n, device="cuda").bfloat16()
    C = torch.zeros(m, n, device="cuda")
    ms = timeit(lambda: ops.gemm_tn(A, B, out=C, accumulate=True))
    print(f"{name:13s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
