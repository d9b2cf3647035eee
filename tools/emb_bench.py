import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = "cuda"
idx = torch.randint(0, 26, (64, 512), device=dev); dout = torch.randn(64, 512, 768, device=dev); dt = torch.zeros(26, 768, device=dev)
for _ in range(3): ops.embedding_bwd(dout, idx, dt)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.embedding_bwd(dout, idx, dt)
e1.record(); torch.cuda.synchronize()
print("embedding_bwd [32768 x 768, V=26]:", e0.elapsed_time(e1) / 20 * 1e3, "us")
