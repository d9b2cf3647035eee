"""Interleaved same-process A/B of GEMM library builds on a few NT shapes: python tools/gemm_ab.py libA.so libB.so ...
   ("-" = the in-tree build).  Each library is loaded side by side through ctypes; rounds alternate between them."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import _lib

libs = sys.argv[1:] or ["-"]
handles = []
for l in libs:
    h = ctypes.CDLL(_lib.LIB_PATH if l == "-" else os.path.abspath(l))
    h.sedd_gemm_nt.restype = ctypes.c_int
    handles.append(h)
P, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
shapes = [(4096, 4096, 4096), (32768, 2304, 768), (32768, 768, 768), (32768, 3072, 768), (32768, 768, 3072), (131072, 2304, 768), (131072, 768, 768)]
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    if os.environ.get("AB_DATA") == "zeros": A.zero_(); B.zero_()       # (power experiment: no operand toggling)
    if os.environ.get("AB_DATA") == "ones": A.fill_(1.0); B.fill_(1.0)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def run(h):
        rc = h.sedd_gemm_nt(P(A.data_ptr()), L(K), P(B.data_ptr()), L(K), P(C.data_ptr()), L(N), I(1), I(1), P(0), I(0), P(0),
                            I(0), L(M), L(N), L(K), P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    best = [1e9] * len(handles)
    for rnd in range(5):
        for i, h in enumerate(handles):
            for _ in range(2): run(h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(h)
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / 10 * 1e3)
    outs = []
    for h in handles:
        C.zero_(); run(h); outs.append(C.clone())
    diff = max(float((o.float() - outs[0].float()).abs().max()) for o in outs)
    print(f"M={M} N={N} K={K}: [max |dC| between builds {diff:g}] " + "  ".join(f"{libs[i]}: {best[i]:.1f} us ({2*M*N*K/best[i]/1e6:.0f} TF/s)" for i in range(len(libs))), flush=True)
