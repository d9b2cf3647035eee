"""Shader clock the chip holds inside the big-tile GEMM (diagnostic build ab_libs/libsedd_clock.so, -DGEMM_CLOCK_STAMP):
every block stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its whole persistent loop; after ~2 s of
back-to-back launches on random data the ratio is the in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
h = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_libs", "libsedd_clock.so"))
h.sedd_gemm_nt.restype = ctypes.c_int
P, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
for M, N, K in ((4096, 4096, 4096), (32768, 3072, 768), (32768, 768, 3072)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        assert h.sedd_gemm_nt(P(A.data_ptr()), L(K), P(B.data_ptr()), L(K), P(C.data_ptr()), L(N), I(1), I(1), P(0), I(0), P(0),
                              I(0), L(M), L(N), L(K), P(torch.cuda.current_stream().cuda_stream)) == 0
    for _ in range(50): run()
    torch.cuda.synchronize()
    n = 6000 if K == 4096 else 4000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    buf = (ctypes.c_uint64 * 2048)()
    assert h.sedd_debug_gemm_stamps(buf) == 0
    ratios = sorted(buf[2 * i] / max(buf[2 * i + 1], 1) * 100e6 / 1e9 for i in range(256) if buf[2 * i + 1] > 0)
    med = ratios[len(ratios) // 2]
    tf = 2.0 * M * N * K / us / 1e6
    print(f"M={M} N={N} K={K}: {us:.1f} us {tf:.0f} TF/s after {n} launches; in-kernel clock median {med:.3f} GHz "
          f"(min {ratios[0]:.3f}, max {ratios[-1]:.3f}) -> MFMA peak at that clock {4096*256*med/1e3:.0f} TF/s, achieved {tf/(4096*256*med/1e3)*100:.0f}% of it", flush=True)
