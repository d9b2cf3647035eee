"""GEMM launch time with warm operands (back-to-back launches on the same buffers) vs cold ones (a 1 GiB fill between
launches evicts L2 / MALL), per launch with HIP events: how much of the in-step GEMM rate is cache state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = torch.device("cuda", 0)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
def one(fn, cold, n=8):
    ts = []
    for _ in range(n):
        if cold:
            flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
M = 32768
for name, N, K in (("qkv fwd", 2304, 768), ("out fwd", 768, 768), ("fc1 fwd", 3072, 768), ("fc2 fwd", 768, 3072)):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm_nt(A, B, out=C)
    for _ in range(3): f()
    w, c = one(f, False), one(f, True)
    fl = 2.0 * M * N * K
    print(f"{name}: warm {w:.1f} us ({fl/w/1e6:.0f} TF/s)   cold {c:.1f} us ({fl/c/1e6:.0f} TF/s)", flush=True)
# sustained: 2 s of back-to-back launches, rate over the last second
A = torch.randn(M, 768, device=dev).bfloat16(); B = torch.randn(3072, 768, device=dev).bfloat16()
C = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
for secs in (0.2, 2.0):
    n = int(secs / 150e-6)
    for _ in range(n): ops.gemm_nt(A, B, out=C)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): ops.gemm_nt(A, B, out=C)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 500 * 1e3
    print(f"fc1 fwd after {secs:.1f} s of sustained launches: {t:.1f} us ({2.0*M*3072*768/t/1e6:.0f} TF/s)", flush=True)
