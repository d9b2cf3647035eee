"""Self-attention fwd / bwd kernel times at the bench shape (packed qkv [B, L, 3, H, d], bf16).
   python tools/attn_bench.py [B H L d] ; SEDD_HIP_LIB=<variant .so> for same-device A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops

B, H, L, d = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 12, 512, 64)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
qkv = torch.randn(B, L, 3, H, d, device=dev, dtype=torch.bfloat16)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
do = torch.randn(B, L, H, d, device=dev, dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv)
o, lse = ops.attn_fwd(q, k, v)

def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tf = timed(lambda: ops.attn_fwd(q, k, v, out=o))
tb = timed(lambda: ops.attn_bwd(q, k, v, o, do, lse, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]))
fl = 4.0 * B * H * L * L * d
print(f"{os.environ.get('SEDD_HIP_LIB', 'default')}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF/s)  "
      f"bwd {tb:.1f} us ({2.5 * fl / tb / 1e6:.0f} TF/s)  chk {float(o.float().sum()):.4f} {float(dqkv.float().sum()):.4f}")
