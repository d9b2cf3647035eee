"""A/B of the XCD tile map's column split (SEDD_GEMM_CSPLIT is read once per process: run once per value) on the three
big-output K = 768 launches of a cfg-2 layer and two A-heavy ones; warm and cold (L2 / MALL flushed) operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = torch.device("cuda", 0)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
def t(fn, cold, n=12):
    ts = []
    for _ in range(3): fn()
    for _ in range(n):
        if cold: flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
M = 32768
A768 = torch.randn(M, 768, device=dev).bfloat16()
A3072 = torch.randn(M, 3072, device=dev).bfloat16()
W1 = (0.05 * torch.randn(3072, 768, device=dev)).bfloat16()
W2 = (0.05 * torch.randn(768, 3072, device=dev)).bfloat16()
Wq = (0.05 * torch.randn(2304, 768, device=dev)).bfloat16()
b1 = torch.randn(3072, device=dev)
pre = torch.randn(M, 3072, device=dev).bfloat16()
cs = torch.zeros(3072, device=dev)
L, H, d = 512, 12, 64
cos, sin = torch.randn(L, 32, device=dev), torch.randn(L, 32, device=dev)
cases = (("fc1 fwd gelu+pre", 2.0 * M * 3072 * 768, lambda: ops.gemm_nt(A768, W1, bias=b1, gelu=True, want_pre=True)),
         ("dgelu nn", 2.0 * M * 3072 * 768, lambda: ops.gemm_nn_dgelu(A768, W2, pre, colsum=cs)),
         ("qkv rotary", 2.0 * M * 2304 * 768, lambda: ops.gemm_nt_rotary(A768, Wq, cos, sin, L, H, d)),
         ("fc2 fwd", 2.0 * M * 3072 * 768, lambda: ops.gemm_nt(A3072, W2)),
         ("dh2 nn", 2.0 * M * 3072 * 768, lambda: ops.gemm_nn(A3072, W1)))
print("SEDD_GEMM_CSPLIT =", os.environ.get("SEDD_GEMM_CSPLIT"))
for name, fl, fn in cases:
    w, c = t(fn, False), t(fn, True)
    print(f"{name:18s} warm {w:7.1f} us {fl/w/1e6:6.0f} TF/s   cold {c:7.1f} us {fl/c/1e6:6.0f} TF/s", flush=True)
