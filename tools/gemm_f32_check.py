import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from protein_lig_sedd_amd import ops
from conftest import rel_err
dev = 'cuda'
torch.manual_seed(0)
bad = 0
for (M, N, K) in [(128,128,64),(200,300,136),(8,4608,128),(1024,26,768),(64,64,8),(257,129,776),(2048,1024,1024),(512,1024,4096),(4,6144,128),(300,136,96),(1,4,4)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    ref = A.double() @ B.double().T
    e = [rel_err(ops.gemm_nt(A, B), ref), rel_err(ops.gemm_nt(A, B, bias=bias), ref + bias.double())]
    C = torch.randn(M, N, device=dev); C0 = C.clone(); ops.gemm_nt(A, B, out=C, accumulate=True); e.append(rel_err(C, C0.double() + ref))
    wide = torch.randn(M, K + 64, device=dev)
    e.append(rel_err(ops.gemm_nt(wide[:, 32:32+K], B), wide[:, 32:32+K].double() @ B.double().T))
    e.append(rel_err(ops.gemm_nt(A, B, out_dtype=torch.bfloat16).float(), ref) / 2e3)
    if N % 4 == 0:
        Bk = B.t().contiguous()
        assert ops.gemm_nn_supported(A, Bk)
        e.append(rel_err(ops.gemm_nn(A, Bk), ref)); e.append(rel_err(ops.gemm_nn(A, Bk, bias=bias), ref + bias.double()))
        Bw = torch.zeros(K, N + 8, device=dev); Bw[:, 4:4+N] = Bk
        e.append(rel_err(ops.gemm_nn(A, Bw[:, 4:4+N]), ref))
    if N % 4 == 0 and M % 4 == 0:
        Ak, Bk = A.t().contiguous(), B.t().contiguous()
        assert ops.gemm_tn_supported(Ak, Bk)
        e.append(rel_err(ops.gemm_tn(Ak, Bk), ref))
        C = torch.randn(M, N, device=dev); C0 = C.clone(); ops.gemm_tn(Ak, Bk, out=C, accumulate=True); e.append(rel_err(C, C0.double() + ref))
        C = torch.randn(M, N, device=dev); ops.gemm_tn(Ak, Bk, out=C, accumulate=False); e.append(rel_err(C, ref))
    ok = max(e) < 2e-6
    bad += not ok
    print((M, N, K), "ok" if ok else "FAIL", " ".join(f"{x:.1e}" for x in e), flush=True)
# odd-K TN (tokens not a multiple of anything)
for (K, M, N) in [(7, 64, 64), (2049, 128, 260), (33, 4, 8), (5000, 64, 128)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    ref = A.double().T @ B.double()
    e = rel_err(ops.gemm_tn(A, B), ref); ok = e < 2e-6; bad += not ok
    print("tn", (K, M, N), "ok" if ok else "FAIL", f"{e:.1e}", flush=True)
print("FAILED" if bad else "ALL OK")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K) in [(2048,1024,1024),(2048,3072,1024),(2048,4096,1024),(2048,1024,4096),(512,1024,1024),(512,3072,1024),(512,4096,1024),(512,1024,4096),(8192,4096,1024)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); Bk = B.t().contiguous(); Ak = A.t().contiguous()
    t1 = bench(lambda: ops.gemm_nt(A, B)); t2 = bench(lambda: ops.gemm_nn(A, Bk)); t3 = bench(lambda: ops.gemm_tn(Ak, Bk))
    fl = 2.0 * M * N * K
    print(f"{(M,N,K)}: nt {t1*1e6:.0f} us {fl/t1/1e12:.1f} TF/s | nn {t2*1e6:.0f} us {fl/t2/1e12:.1f} | tn {t3*1e6:.0f} us {fl/t3/1e12:.1f}", flush=True)
