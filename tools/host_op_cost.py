"""Host cost of one op wrapper call (tiny tensors, GPU idle): where the ~11 us per launch of a launch-bound step go."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops, _lib
dev = "cuda"
A = torch.randn(64, 64, device=dev); B = torch.randn(64, 64, device=dev); bias = torch.randn(64, device=dev)
out = torch.empty(64, 64, device=dev)
L = _lib.lib()
def t(f, n=20000):
    for _ in range(200): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print(f"ops.gemm_nt(A, B, bias)            {t(lambda: ops.gemm_nt(A, B, bias=bias)):.2f} us")
print(f"ops.gemm_nt(A, B, out=out)         {t(lambda: ops.gemm_nt(A, B, out=out)):.2f} us")
pa, pb, po, st = A.data_ptr(), B.data_ptr(), out.data_ptr(), ops._stream()
print(f"raw ctypes sedd_gemm_nt            {t(lambda: L.sedd_gemm_nt(pa, 64, pb, 64, po, 64, 0, 0, None, 0, None, 0, 64, 64, 64, st)):.2f} us")
print(f"torch.empty(64, 64)                {t(lambda: torch.empty(64, 64, dtype=torch.float32, device=A.device)):.2f} us")
print(f"ops._stream()                      {t(ops._stream):.2f} us")
print(f"A.data_ptr() x3 + stride x3        {t(lambda: (A.data_ptr(), B.data_ptr(), out.data_ptr(), A.stride(0), B.stride(0), out.stride(0))):.2f} us")
print(f"with ops._Timed(...)               {t(lambda: ops._Timed('x', 1.0).__enter__().__exit__()):.2f} us")
x = torch.randn(4, 16, 64, device=dev)
print(f"x.view + slice                     {t(lambda: x.view(64, 64)[:, :32]):.2f} us")
print(f"torch.add (aten op)                {t(lambda: torch.add(A, B, out=out)):.2f} us")
print(f"ops.cast(A, bf16)                  {t(lambda: ops.cast(A, torch.bfloat16)):.2f} us")
