"""embedding gradient, owner-computes shapes of the cross model (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = "cuda"
for V, M, n in ((int(sys.argv[1]), int(sys.argv[2]), 5),):
    D = 1024
    idx = torch.randint(0, V - 1, (M,), device=dev)
    idx[torch.rand(M, device=dev) < 0.5] = V - 1  # (absorbing graph: half the tokens are the mask token)
    dout = torch.randn(M, D, device=dev)
    table = torch.zeros(V, D, device=dev)
    for _ in range(n):
        ops.embedding_bwd(dout, idx, table)
    torch.cuda.synchronize()
    ref = torch.zeros(V, D, device=dev).index_add_(0, idx, dout) * n
    print(V, M, float((table - ref).abs().max()))
