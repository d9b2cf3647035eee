"""What happens to the persistent one-block-per-CU GEMM while another kernel holds some CUs (a stand-in for a resident
RCCL collective): time of the fc1-forward GEMM alone and with `nwg` occupying workgroups resident on a second stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libprobe.so"))
P.probe_occupy.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
M, N, K = 32768, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16(); B = (0.05 * torch.randn(N, K, device=dev)).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream()
def gemm_time(n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm_nt(A, B, out=C)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for budget in (256, 248, 240):
  ops.set_gemm_cu_budget(budget)
  for _ in range(5): ops.gemm_nt(A, B, out=C)
  torch.cuda.synchronize()
  print(f"CU budget {budget}: alone {gemm_time():.1f} us per launch", flush=True)
  for nwg in (1, 8, 16, 64):
    if nwg > 256 - budget and budget != 256: continue
    # occupy for 6 ms (600000 ticks of 10 ns); give the occupying kernel 0.5 ms to become resident, then time 10 GEMMs
    P.probe_occupy(nwg, 600000, sink.data_ptr(), side.cuda_stream)
    import time; time.sleep(0.0005)
    t = gemm_time()
    torch.cuda.synchronize()
    print(f"   {nwg:3d} occupying workgroups resident: {t:.1f} us per launch", flush=True)
