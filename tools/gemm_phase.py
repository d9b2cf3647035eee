"""Where a persistent GEMM block spends its time (diagnostic builds ab_libs/libsedd_phase*.so, -DGEMM_PHASE_STAMP): every
block stamps s_memrealtime (100 MHz) at the top of each tile round, after the last MFMA group (ring free) and after the
epilogue's last store has been ISSUED.  Prints, per shape, the median main-loop and epilogue time per round over the first
256 blocks, and the spread of the round start times (how much in step the blocks run).
usage: gemm_phase.py [lib] [fc1|plain]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "ab_libs", "libsedd_phase.so")
h = ctypes.CDLL(lib)
h.sedd_gemm_nt.restype = ctypes.c_int
P, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
for M, N, K, epi, want_pre in ((32768, 3072, 768, 2, 1), (32768, 3072, 768, 2, 0), (32768, 3072, 768, 1, 0), (32768, 3072, 768, 0, 0),
                             (32768, 768, 3072, 0, 0), (4096, 4096, 4096, 0, 0)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if want_pre else None
    bias = torch.randn(N, device="cuda") if epi else None
    def run():
        assert h.sedd_gemm_nt(P(A.data_ptr()), L(K), P(B.data_ptr()), L(K), P(C.data_ptr()), L(N), I(1), I(1),
                              P(bias.data_ptr() if epi else 0), I(epi), P(pre.data_ptr() if pre is not None else 0),
                              I(0), L(M), L(N), L(K), P(torch.cuda.current_stream().cuda_stream)) == 0
    for _ in range(20): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    NB = 512 if os.environ.get('SEDD_GEMM_NW4') else 256
    buf = (ctypes.c_uint64 * (512 * 8 * 3))()
    assert h.sedd_debug_gemm_phase(buf) == 0
    rounds = min(8, (M // 256) * (N // (128 if NB == 512 else 256)) // NB)
    t00 = min(buf[(b * 8) * 3] for b in range(NB))
    print(f"M={M} N={N} K={K} epi={epi} pre={want_pre}: {us:.1f} us, {2.0*M*N*K/us/1e6:.0f} TF/s, {rounds} rounds", flush=True)
    for r, lo in ((r, lo) for r in range(rounds) for lo in range(0, NB, 256)):
        st = sorted((buf[(b * 8 + r) * 3] - t00) / 100.0 for b in range(lo, lo + 256))
        ml = sorted((buf[(b * 8 + r) * 3 + 1] - buf[(b * 8 + r) * 3]) / 100.0 for b in range(lo, lo + 256))
        ep = sorted((buf[(b * 8 + r) * 3 + 2] - buf[(b * 8 + r) * 3 + 1]) / 100.0 for b in range(lo, lo + 256))
        print(f"  round {r} blocks {lo}+: start {st[0]:.1f}..{st[128]:.1f}..{st[-1]:.1f} us   main loop {ml[0]:.1f}/{ml[128]:.1f}/{ml[-1]:.1f} us"
              f"   epilogue {ep[0]:.1f}/{ep[128]:.1f}/{ep[-1]:.1f} us (min/median/max)", flush=True)

# the fused dGELU data gradient (dX = (dY W2) * gelu'(u): M x 3072, K = 768, B as [K, N]) and the QKV projection with rotary
def phase_report(tag, us, flops, NB_, tiles):
    buf = (ctypes.c_uint64 * (512 * 8 * 3))()
    assert h.sedd_debug_gemm_phase(buf) == 0
    rounds = min(8, max(1, tiles // NB_))
    print(f"{tag}: {us:.1f} us, {flops/us/1e6:.0f} TF/s, {rounds} rounds", flush=True)
    for r in range(rounds):
        ml = sorted((buf[(b * 8 + r) * 3 + 1] - buf[(b * 8 + r) * 3]) / 100.0 for b in range(256))
        ep = sorted((buf[(b * 8 + r) * 3 + 2] - buf[(b * 8 + r) * 3 + 1]) / 100.0 for b in range(256))
        print(f"  round {r}: main loop {ml[0]:.1f}/{ml[128]:.1f}/{ml[-1]:.1f} us   epilogue {ep[0]:.1f}/{ep[128]:.1f}/{ep[-1]:.1f} us", flush=True)

def bench(fn):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3

if NB == 256:
    M, N, K = 32768, 3072, 768
    dY = torch.randn(M, K, device="cuda").bfloat16(); W2 = torch.randn(K, N, device="cuda").bfloat16()
    pre = torch.randn(M, N, device="cuda").bfloat16(); C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cs = torch.zeros(N, device="cuda")
    h.sedd_gemm_nn_dgelu.restype = ctypes.c_int
    def run_dg():
        assert h.sedd_gemm_nn_dgelu(P(dY.data_ptr()), L(K), P(W2.data_ptr()), L(N), P(C.data_ptr()), L(N), I(1), I(1),
                                    P(pre.data_ptr()), P(cs.data_ptr()), L(M), L(N), L(K), P(torch.cuda.current_stream().cuda_stream)) == 0
    phase_report("dGELU NN M=32768 N=3072 K=768", bench(run_dg), 2.0 * M * N * K, 256, (M // 256) * (N // 256))
    N = 2304
    X = torch.randn(M, K, device="cuda").bfloat16(); Wq = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cos = torch.rand(512, 32, device="cuda"); sin = torch.rand(512, 32, device="cuda")
    h.sedd_gemm_nt_rotary.restype = ctypes.c_int
    def run_rot():
        assert h.sedd_gemm_nt_rotary(P(X.data_ptr()), L(K), P(Wq.data_ptr()), L(K), P(C.data_ptr()), L(N), I(1), I(1), P(0), I(0),
                                     P(cos.data_ptr()), P(sin.data_ptr()), I(512), I(12), I(64), L(M), L(N), L(K),
                                     P(torch.cuda.current_stream().cuda_stream)) == 0
    phase_report("QKV rotary M=32768 N=2304 K=768 (256 x 128 tiles)", bench(run_rot), 2.0 * M * N * K, 256, (M // 256) * (N // 128))
    def run_plain():
        assert h.sedd_gemm_nt(P(X.data_ptr()), L(K), P(Wq.data_ptr()), L(K), P(C.data_ptr()), L(N), I(1), I(1), P(0), I(0), P(0),
                              I(0), L(M), L(N), L(K), P(torch.cuda.current_stream().cuda_stream)) == 0
    phase_report("QKV plain  M=32768 N=2304 K=768", bench(run_plain), 2.0 * M * N * K, 256, (M // 256) * (N // 256))
