"""Weight-gradient (TN) GEMM on the cfg-2 / cfg-4 shapes, back to back and with the step's other traffic between launches
(a 400 MB copy evicts the Infinity Cache): microseconds and TF/s incl. the split-K fold.  SEDD_TN_WIDE=0|1 forces the tile width."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops

def timeit(fn, n=20, evict=None):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if evict is not None: evict()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n

big = torch.empty(100 << 20, device="cuda", dtype=torch.float32)
big2 = torch.empty_like(big)
for name, m, n, k in (("fc1 wgrad", 3072, 768, 32768), ("fc2 wgrad", 768, 3072, 32768), ("qkv wgrad", 2304, 768, 32768),
                      ("out wgrad", 768, 768, 32768), ("cfg4 fc1", 4096, 1024, 16384), ("cfg4 out", 1024, 1024, 16384)):
    A = torch.randn(k, m, device="cuda").bfloat16(); B = torch.randn(k, n, device="cuda").bfloat16()
    C = torch.zeros(m, n, device="cuda")
    warm = timeit(lambda: ops.gemm_tn(A, B, out=C, accumulate=True))
    cold = timeit(lambda: ops.gemm_tn(A, B, out=C, accumulate=True), evict=lambda: big2.copy_(big))
    print(f"{name:10s} M={m} N={n} K={k}: warm {warm*1e3:7.1f} us {2*m*n*k/warm/1e9:7.1f} TF/s   cold {cold*1e3:7.1f} us {2*m*n*k/cold/1e9:7.1f} TF/s", flush=True)
