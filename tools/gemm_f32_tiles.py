"""fp32 GEMM time per tile shape (SEDD_F32_TILE = 0: 128x128, 1: 128x64, 2: 64x64) — run once per value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
dev = "cuda"
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
out = []
for (M, N, K) in [(2048,1024,1024),(2048,2048,1024),(2048,3072,1024),(2048,4096,1024),(2048,1024,4096),(512,1024,1024),(512,2048,1024),(512,3072,1024),(512,4096,1024),(512,1024,4096),(1024,1024,2048),(1024,4096,2048),(1024,1024,512),(4096,1024,512),(8192,4096,1024)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
    t = bench(lambda: ops.gemm_nt(A, B))
    out.append(f"{M}x{N}x{K}:{t*1e6:.0f}us/{2.0*M*N*K/t/1e12:.0f}TF")
print("tile", os.environ.get("SEDD_F32_TILE"), " ".join(out))
