"""GPU parity of the model kernels (through the C ABI) against plain fp32 torch references of the same op and the CPU
oracle (oracle/sedd_cpu.py).  Tolerances: fp32 kernels 1e-4 relative (north_star), bf16 kernels 2e-2."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, rel_err
from oracle import sedd_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from protein_lig_sedd_amd import ops
    return ops


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 300, 136), (8, 4608, 128), (1024, 26, 768), (64, 64, 8),
                                   (257, 129, 776), (512, 1024, 2048), (2048, 1024, 1024), (1, 4, 4)])
def test_gemm_fp32(M, N, K):
    ops = _ops()
    torch.manual_seed(M + N + K)
    A, B = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
    bias = torch.randn(N, device=DEV)
    ref = A.double() @ B.double().T
    assert rel_err(ops.gemm_nt(A, B), ref) < 2e-6
    assert rel_err(ops.gemm_nt(A, B, bias=bias), ref + bias.double()) < 2e-6
    out, pre = ops.gemm_nt(A, B, bias=bias, gelu=True, want_pre=True)
    assert rel_err(pre, ref + bias.double()) < 2e-6
    assert rel_err(out, F.gelu((ref + bias.double()).float(), approximate="tanh")) < 1e-5
    C = torch.randn(M, N, device=DEV)
    C0 = C.clone()
    ops.gemm_nt(A, B, out=C, accumulate=True)
    assert rel_err(C, C0.double() + ref) < 2e-6
    assert rel_err(ops.gemm_nt(A, B, bias=bias, out_dtype=torch.bfloat16).float(), ref + bias.double()) < 4e-3  # bf16 result
    wide = torch.randn(M, K + 8, device=DEV)
    assert rel_err(ops.gemm_nt(wide[:, 4:4 + K], B), wide[:, 4:4 + K].double() @ B.double().T) < 2e-6  # row stride > K


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 300, 136), (8, 4608, 128), (64, 64, 8), (512, 1024, 2048),
                                   (2048, 1024, 1024), (300, 136, 96), (1, 4, 4), (1028, 260, 72)])
def test_gemm_nn_tn_fp32(M, N, K):
    """the exact-fp32 kernel's other two operand layouts — B stored [K, N] (data gradient), A stored [K, M] and B [K, N]
    (weight gradient, any K) — against float64, incl. strided operands, bias, accumulation and the split-K shapes."""
    ops = _ops()
    torch.manual_seed(M + N + K)
    A, B = torch.randn(M, K, device=DEV), torch.randn(K, N, device=DEV)
    bias = torch.randn(N, device=DEV)
    ref = A.double() @ B.double()
    assert ops.gemm_nn_supported(A, B)
    assert rel_err(ops.gemm_nn(A, B), ref) < 2e-6
    assert rel_err(ops.gemm_nn(A, B, bias=bias), ref + bias.double()) < 2e-6
    Aw = torch.randn(M, K + 8, device=DEV)
    Bw = torch.randn(K, N + 8, device=DEV)
    out = torch.full((M, N + 4), 7.0, device=DEV)
    ops.gemm_nn(Aw[:, 4:4 + K], Bw[:, 4:4 + N], out=out[:, :N])
    assert rel_err(out[:, :N], Aw[:, 4:4 + K].double() @ Bw[:, 4:4 + N].double()) < 2e-6
    assert bool((out[:, N:] == 7.0).all())
    if M % 4 == 0:
        At = A.t().contiguous()  # [K, M]
        assert ops.gemm_tn_supported(At, B)
        assert rel_err(ops.gemm_tn(At, B), ref) < 2e-6
        C = torch.randn(M, N, device=DEV)
        C0 = C.clone()
        ops.gemm_tn(At, B, out=C, accumulate=True)
        assert rel_err(C, C0.double() + ref) < 2e-6
        ops.gemm_tn(At, B, out=C, accumulate=False)
        assert rel_err(C, ref) < 2e-6
        odd = At[:K - 1] if K > 1 else At  # a reduction length that is a multiple of nothing
        assert rel_err(ops.gemm_tn(odd, B[:odd.shape[0]]), odd.double().T @ B[:odd.shape[0]].double()) < 2e-6
    assert not ops.gemm_nn_supported(A, torch.randn(K, N + 1, device=DEV)[:, :N + 1])
    assert not ops.gemm_tn_supported(torch.randn(K, M + 2, device=DEV), B)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 300, 136), (64, 4608, 128), (2048, 26, 768), (512, 768, 3072),
                                   (257, 129, 776), (256, 256, 32), (300, 200, 96), (1000, 130, 160), (777, 515, 224),
                                   (768, 2304, 8192), (192, 96, 64)])
def test_gemm_bf16(M, N, K):
    ops = _ops()
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=DEV).bfloat16()
    B = torch.randn(N, K, device=DEV).bfloat16()
    bias = torch.randn(N, device=DEV)
    ref = A.double() @ B.double().T  # exact products of the bf16 inputs
    out32 = ops.gemm_nt(A, B, out_dtype=torch.float32)
    assert rel_err(out32, ref) < 1e-5  # fp32 accumulation of exact bf16 products
    out = ops.gemm_nt(A, B, bias=bias)
    assert out.dtype == torch.bfloat16
    assert rel_err(out, ref + bias.double()) < 5e-3
    g, pre = ops.gemm_nt(A, B, bias=bias, gelu=True, want_pre=True)
    assert rel_err(pre, ref + bias.double()) < 5e-3
    assert rel_err(g, F.gelu(pre.float(), approximate="tanh")) < 5e-3
    C = torch.randn(M, N, device=DEV)
    C0 = C.clone()
    ops.gemm_nt(A, B, out=C, accumulate=True)
    assert rel_err(C, C0.double() + ref) < 1e-5
    # strided A (a column slab of a wider matrix) as used for packed activations
    wide = torch.randn(M, K + 64, device=DEV).bfloat16()
    assert rel_err(ops.gemm_nt(wide[:, 32:32 + K], B, out_dtype=torch.float32),
                   wide[:, 32:32 + K].double() @ B.double().T) < 1e-5


@pytest.mark.parametrize("K,M,N", [(32, 256, 256), (64, 96, 128), (4096, 768, 3072), (1024, 2304, 768), (160, 200, 136),
                                   (8192, 776, 520), (96, 8, 8)])
def test_gemm_tn_bf16(K, M, N):
    """weight-gradient GEMM C = A^T B from row-major [tokens, features] operands (transposing LDS reads, split-K)."""
    ops = _ops()
    torch.manual_seed(K + M + N)
    A = torch.randn(K, M, device=DEV).bfloat16()
    B = torch.randn(K, N, device=DEV).bfloat16()
    assert ops.gemm_tn_supported(A, B)
    ref = A.double().T @ B.double()
    assert rel_err(ops.gemm_tn(A, B), ref) < 1e-5
    assert rel_err(ops.gemm_tn(A, B, out_dtype=torch.bfloat16), ref) < 5e-3
    C = torch.randn(M, N, device=DEV)
    C0 = C.clone()
    ops.gemm_tn(A, B, out=C, accumulate=True)
    assert rel_err(C, C0.double() + ref) < 1e-5
    C = torch.randn(M, N, device=DEV)
    ops.gemm_tn(A, B, out=C, accumulate=False)
    assert rel_err(C, ref) < 1e-5
    # column slabs of wider matrices (packed activations)
    wa = torch.randn(K, M + 64, device=DEV).bfloat16()
    wb = torch.randn(K, N + 32, device=DEV).bfloat16()
    assert rel_err(ops.gemm_tn(wa[:, 32:32 + M], wb[:, 8:8 + N]), wa[:, 32:32 + M].double().T @ wb[:, 8:8 + N].double()) < 1e-5
    assert not ops.gemm_tn_supported(A[:31], B[:31])


@pytest.mark.parametrize("K,M,N", [(512, 1024, 1024), (2048, 3072, 1024), (2048, 1024, 4096), (4096, 1024, 1024),
                                   (32768, 768, 3072), (96, 264, 136), (2048, 128, 1024), (8192, 40, 72)])
def test_gemm_tn_colsum(K, M, N):
    """weight gradient with the bias gradient (column sums of dY) out of the same launch: unsplit short reductions,
    split ones (partial sums behind the partial tiles), ragged edges, accumulate on and off, repeatable bit for bit."""
    ops = _ops()
    torch.manual_seed(K + M + N)
    A = torch.randn(K, M, device=DEV).bfloat16()
    B = torch.randn(K, N, device=DEV).bfloat16()
    ref, cref = A.double().T @ B.double(), A.double().sum(0)
    cs = torch.full((M,), 7.0, device=DEV)
    C = ops.gemm_tn(A, B, colsum=cs)                       # fresh C: both are overwritten
    assert rel_err(C, ref) < 1e-5 and rel_err(cs, cref) < 1e-5
    C1 = torch.randn(M, N, device=DEV)
    c1 = torch.randn(M, device=DEV)
    C0, c0 = C1.clone(), c1.clone()
    ops.gemm_tn(A, B, out=C1, accumulate=True, colsum=c1)
    assert rel_err(C1, C0.double() + ref) < 1e-5 and rel_err(c1, c0.double() + cref) < 1e-5
    C2, c2 = C0.clone(), c0.clone()
    ops.gemm_tn(A, B, out=C2, accumulate=True, colsum=c2)
    assert torch.equal(C1, C2) and torch.equal(c1, c2)
    # the product itself is the one the plain entry point gives when both take the same tiles (unsplit / 128-wide)
    wa = torch.randn(K, M + 64, device=DEV).bfloat16()
    c3 = torch.zeros(M, device=DEV)
    ops.gemm_tn(wa[:, 32:32 + M], B, out=torch.zeros(M, N, device=DEV), accumulate=True, colsum=c3)
    assert rel_err(c3, wa[:, 32:32 + M].double().sum(0)) < 1e-5
    with pytest.raises(ValueError):
        ops.gemm_tn(A.float(), B.float(), colsum=cs)


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 3072, 768), (300, 200, 96)])
def test_gemm_nt_dgelu(M, N, K):
    """fused data gradient through GELU + bias-gradient column sums == gemm -> gelu_bwd -> colsum."""
    ops = _ops()
    torch.manual_seed(M + N + K)
    A = (0.3 * torch.randn(M, K, device=DEV)).bfloat16()
    B = (0.3 * torch.randn(N, K, device=DEV)).bfloat16()
    pre = torch.randn(M, N, device=DEV).bfloat16()
    assert ops.gemm_nt_dgelu_supported(A, B, pre)
    cs = torch.ones(N, device=DEV)
    out = ops.gemm_nt_dgelu(A, B, pre, colsum=cs)
    ref = ops.gelu_bwd(pre, ops.gemm_nt(A, B))
    assert rel_err(out.float(), ref.float()) < 2e-3  # (hardware-exp tanh in the epilogue vs tanhf: bf16 rounding flips)
    assert rel_err(cs, 1 + ref.float().sum(0)) < 1e-4
    pf = pre.float().requires_grad_(True)
    F.gelu(pf, approximate="tanh").backward((A.double() @ B.double().T).float())
    assert rel_err(out.float(), pf.grad) < 1e-2


@pytest.mark.parametrize("M,N,K", [(512, 768, 2304), (4096, 768, 768), (300, 136, 96), (1024, 3072, 768), (256, 1000, 64)])
def test_gemm_nn_bf16(M, N, K):
    """C = A[M,K] @ B[K,N] with B as stored (data gradient dY @ W) == the NT GEMM on a transposed copy of B."""
    ops = _ops()
    torch.manual_seed(M + N + K)
    A = (0.3 * torch.randn(M, K, device=DEV)).bfloat16()
    B = (0.3 * torch.randn(K, N, device=DEV)).bfloat16()
    assert ops.gemm_nn_supported(A, B)
    out = ops.gemm_nn(A, B)
    ref = ops.gemm_nt(A, B.t().contiguous())
    assert rel_err(out.float(), ref.float()) < 2e-3  # (same products, another summation order over k)
    assert rel_err(out.float(), (A.double() @ B.double()).float()) < 1e-2
    bias = torch.randn(N, device=DEV)
    assert rel_err(ops.gemm_nn(A, B, bias=bias).float(), ref.float() + bias) < 1e-2
    # strided operands (row stride > extent)
    Ab = torch.zeros(M, K + 64, device=DEV).bfloat16(); Ab[:, :K] = A
    Bb = torch.zeros(K, N + 8, device=DEV).bfloat16(); Bb[:, :N] = B
    assert torch.equal(ops.gemm_nn(Ab[:, :K], Bb[:, :N]), out)


@pytest.mark.parametrize("M,N,K", [(512, 3072, 768), (1024, 384, 96), (320, 1000, 64)])
def test_gemm_nn_dgelu(M, N, K):
    ops = _ops()
    torch.manual_seed(M + N + K + 1)
    A = (0.3 * torch.randn(M, K, device=DEV)).bfloat16()
    B = (0.3 * torch.randn(K, N, device=DEV)).bfloat16()
    pre = torch.randn(M, N, device=DEV).bfloat16()
    cs = torch.ones(N, device=DEV)
    out = ops.gemm_nn_dgelu(A, B, pre, colsum=cs)
    cs2 = torch.ones(N, device=DEV)
    ref = ops.gemm_nt_dgelu(A, B.t().contiguous(), pre, colsum=cs2)
    assert rel_err(out.float(), ref.float()) < 2e-3
    assert rel_err(cs, cs2) < 1e-3


@pytest.mark.parametrize("M,N,K", [(512, 3072, 768), (1000, 384, 96), (320, 1000, 64), (200, 136, 72), (16, 128, 64)])
def test_gemm_gelu_grad_epilogue_and_dmul(M, N, K):
    """forward epilogue that stores gelu'(pre) next to gelu(pre) (SEDD_EPI_BIAS_GELU_GRAD, every kernel family: big-tile
    staged, small-tile, generic-K) and the backward's multiply epilogue (sedd_gemm_nn_dmul / _nt_dmul): the pair must
    reproduce Linear -> GELU(tanh)'s autograd (transformer.py:129-133) like the pre-activation / dGELU pair does."""
    ops = _ops()
    torch.manual_seed(M + N + K + 2)
    X = (0.5 * torch.randn(M, K, device=DEV)).bfloat16()
    W1 = (0.3 * torch.randn(N, K, device=DEV)).bfloat16()
    b1 = torch.randn(N, device=DEV)
    g0, u = ops.gemm_nt(X, W1, bias=b1, gelu=True, want_pre=True)
    g1, gp = ops.gemm_nt(X, W1, bias=b1, gelu=True, want_pre="grad")
    assert torch.equal(g0, g1)
    uf = u.float().requires_grad_(True)
    F.gelu(uf, approximate="tanh").sum().backward()
    assert rel_err(gp.float(), uf.grad) < 4e-3          # bf16 rounding of the stored derivative
    assert float((gp.float() - uf.grad).abs().max()) < 1e-2
    # fp32 operands: exact tanh forms
    g32, gp32 = ops.gemm_nt(X.float(), W1.float(), bias=b1, gelu=True, want_pre="grad")
    u32 = (X.float().double() @ W1.float().double().T + b1.double()).float().requires_grad_(True)
    F.gelu(u32, approximate="tanh").sum().backward()
    assert rel_err(gp32, u32.grad) < 1e-5 and rel_err(g32, F.gelu(u32.detach(), approximate="tanh")) < 1e-5
    if not ops.gemm_dmul_shapes_ok(M, N, 64, torch.bfloat16):
        return
    Ko = 96
    dY = (0.3 * torch.randn(M, Ko, device=DEV)).bfloat16()
    W2 = (0.3 * torch.randn(Ko, N, device=DEV)).bfloat16()   # second Linear's weight as stored [out, hidden]
    cs, cs_ref = torch.ones(N, device=DEV), torch.ones(N, device=DEV)
    du = ops.gemm_dmul(dY, W2, gp, colsum=cs)
    ref = ops.gemm_nn_dgelu(dY, W2, u, colsum=cs_ref)
    assert rel_err(du.float(), ref.float()) < 6e-3       # one more bf16 rounding (the stored derivative) than the reference
    assert rel_err(cs, cs_ref) < 5e-3
    exact = (dY.double() @ W2.double()).float() * uf.grad
    assert rel_err(du.float(), exact) < 1e-2
    assert rel_err(cs - 1, du.float().sum(0)) < 1e-4     # the column sums are those of the stored result
    du_nt = ops.gemm_dmul(dY, W2.t().contiguous(), gp, b_kn=False)
    assert rel_err(du_nt.float(), du.float()) < 2e-3


def test_gemm_nn_rejects_unsupported():
    ops = _ops()
    from protein_lig_sedd_amd._lib import SeddError
    A = torch.zeros(512, 32, device=DEV).bfloat16()
    assert not ops.gemm_nn_supported(A, torch.zeros(32, 26, device=DEV).bfloat16())
    with pytest.raises(SeddError):
        ops.gemm_nn(A, torch.zeros(32, 104, device=DEV).bfloat16()[:, :100])


@pytest.mark.parametrize("Bt,L,H,K", [(4, 128, 12, 768), (2, 100, 3, 96), (64, 512, 12, 768)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_gemm_nt_rotary_matches_gemm_then_rotary(Bt, L, H, K, with_bias):
    """QKV projection with rotary in the epilogue == sedd_gemm_nt followed by sedd_rotary_qkv (transformer.py:165-171)."""
    ops = _ops()
    d, M = 64, Bt * L
    torch.manual_seed(L + H)
    A = (0.5 * torch.randn(M, K, device=DEV)).bfloat16()
    W = (0.2 * torch.randn(3 * H * d, K, device=DEV)).bfloat16()
    bias = torch.randn(3 * H * d, device=DEV) if with_bias else None
    ang = torch.rand(L, d // 2, device=DEV) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    assert ops.gemm_nt_rotary_supported(A, W, L, H, d)
    got = ops.gemm_nt_rotary(A, W, cos, sin, L, H, d, bias=bias).view(Bt, L, 3, H, d)
    ref = ops.gemm_nt(A, W, bias=bias).view(Bt, L, 3, H, d).clone()
    assert torch.equal(got[:, :, 2], ref[:, :, 2])            # V passes through
    ops.rotary_qkv_(ref, cos, sin)
    # same fp32 arithmetic on the same bf16-rounded projection; fma contraction may move the last bf16 bit
    assert rel_err(got.float(), ref.float()) < 4e-3
    assert float((got.float() - ref.float()).abs().max()) <= 2 ** -6 * float(ref.float().abs().max())


def test_gemm_rejects_bad_shapes():
    ops = _ops()
    from protein_lig_sedd_amd._lib import SeddError
    with pytest.raises(SeddError):
        ops.gemm_nt(torch.zeros(4, 12, device=DEV).bfloat16(), torch.zeros(4, 12, device=DEV).bfloat16())
    with pytest.raises(ValueError):
        ops.gemm_nt(torch.zeros(4, 16, device=DEV), torch.zeros(4, 8, device=DEV))


@pytest.mark.parametrize("B,L,D", [(3, 40, 768), (2, 33, 1024), (2, 5, 64), (1, 128, 1536)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ln_modulate(B, L, D, dtype):
    ops = _ops()
    torch.manual_seed(D + L)
    x = (torch.randn(B, L, D, device=DEV) * 2 + 0.5).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(D, device=DEV)).requires_grad_(True)
    mod = (0.3 * torch.randn(B, 6 * D, device=DEV)).to(dtype)
    modr = mod.float().requires_grad_(True)
    shift, scale = modr[:, :D], modr[:, D:2 * D]
    ref = F.layer_norm(x, [D]) * w * (1 + scale[:, None]) + shift[:, None]
    y, mean, rstd = ops.ln_modulate_fwd(x.detach(), w.detach(), shift=mod[:, :D], scale=mod[:, D:2 * D], out_dtype=dtype)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert rel_err(y, ref) < tol
    dy = torch.randn(B, L, D, device=DEV).to(dtype)
    ref.backward(dy.float())
    dx = torch.randn(B, L, D, device=DEV)
    dx0 = dx.clone()
    dw = torch.zeros(D, device=DEV)
    dmod = torch.zeros(B, 6 * D, device=DEV)
    ops.ln_modulate_bwd(dy, x.detach(), w.detach(), mod[:, D:2 * D], mean, rstd, dx, True, dw, dmod[:, :D], dmod[:, D:2 * D])
    assert rel_err(dx - dx0, x.grad) < 1e-4
    assert rel_err(dw, w.grad) < 1e-4
    assert rel_err(dmod[:, :2 * D], modr.grad[:, :2 * D]) < 1e-4
    assert float(dmod[:, 2 * D:].abs().max()) == 0.0
    # plain affine LayerNorm (cross-attention norm)
    b = torch.randn(D, device=DEV)
    y2, _, _ = ops.ln_modulate_fwd(x.detach(), w.detach(), bias=b)
    assert rel_err(y2, F.layer_norm(x.detach(), [D], w.detach(), b)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gate_residual(dtype):
    ops = _ops()
    torch.manual_seed(3)
    B, L, D = 3, 37, 768
    res = torch.randn(B, L, D, device=DEV, requires_grad=True)
    a = torch.randn(B, L, D, device=DEV).to(dtype)
    ar = a.float().requires_grad_(True)
    mod = torch.randn(B, 6 * D, device=DEV).to(dtype)
    gr = mod[:, 2 * D:3 * D].float().requires_grad_(True)
    keep = (torch.rand(B, L, D, device=DEV) > 0.1).to(torch.uint8)
    inv = 1 / 0.9
    for kp in (None, keep):
        res.grad = ar.grad = gr.grad = None
        ref = res + gr[:, None] * (ar * (kp.float() * inv if kp is not None else 1.0))
        out = ops.gate_residual_fwd(res.detach(), a, mod[:, 2 * D:3 * D], kp, inv)
        assert rel_err(out, ref) < 1e-6
        dout = torch.randn(B, L, D, device=DEV)
        ref.backward(dout)
        dmod = torch.zeros(B, 6 * D, device=DEV)
        da = ops.gate_residual_bwd(dout, a, mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], kp, inv)
        assert rel_err(da, ar.grad) < (1e-6 if dtype == torch.float32 else 5e-3)
        assert rel_err(dmod[:, 2 * D:3 * D], gr.grad) < 1e-4
        # bias-gradient column sums folded into the same pass: dbias += da.sum(rows), da as stored
        dmod2, dbias = torch.zeros(B, 6 * D, device=DEV), torch.ones(D, device=DEV)
        da2 = ops.gate_residual_bwd(dout, a, mod[:, 2 * D:3 * D], dmod2[:, 2 * D:3 * D], kp, inv, dbias=dbias)
        assert torch.equal(da2, da) and torch.equal(dmod2, dmod)
        assert rel_err(dbias, 1 + da.float().sum((0, 1))) < 1e-5


@pytest.mark.parametrize("B,L,D", [(3, 40, 768), (2, 33, 1024), (2, 5, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gate_residual_ln_modulate_fused_matches_the_two_kernels(B, L, D, dtype, p):
    ops = _ops()
    torch.manual_seed(D + L)
    res = torch.randn(B, L, D, device=DEV)
    a = torch.randn(B, L, D, device=DEV).to(dtype)
    w = 1 + 0.1 * torch.randn(D, device=DEV)
    mod = (0.3 * torch.randn(B, 6 * D, device=DEV)).to(dtype)
    keep = ops.DropSeed(p, seed=1234) if p > 0 else None
    inv = 1.0 / (1.0 - p)
    assert ops.gate_residual_ln_modulate_supported(res, a, keep)
    x1, h, mean, rstd = ops.gate_residual_ln_modulate_fwd(res, a, mod[:, 2 * D:3 * D], keep, inv, w, mod[:, 3 * D:4 * D],
                                                           mod[:, 4 * D:5 * D])
    x1r = ops.gate_residual_fwd(res, a, mod[:, 2 * D:3 * D], keep, inv)
    hr, meanr, rstdr = ops.ln_modulate_fwd(x1r, w, shift=mod[:, 3 * D:4 * D], scale=mod[:, 4 * D:5 * D], out_dtype=dtype)
    assert torch.equal(x1, x1r)
    assert torch.equal(mean, meanr) and torch.equal(rstd, rstdr) and torch.equal(h, hr)


@pytest.mark.parametrize("B,L,D", [(3, 40, 768), (2, 70, 1024), (2, 5, 64), (40, 512, 768)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_ln_modulate_bwd_gate_fused_matches_the_two_kernels(B, L, D, dtype, p):
    ops = _ops()
    torch.manual_seed(D + L + 1)
    x = torch.randn(B, L, D, device=DEV) * 2 + 0.5
    dy = torch.randn(B, L, D, device=DEV).to(dtype)
    a = torch.randn(B, L, D, device=DEV).to(dtype)
    add = torch.randn(B, L, D, device=DEV)
    w = 1 + 0.1 * torch.randn(D, device=DEV)
    mod = (0.3 * torch.randn(B, 6 * D, device=DEV)).to(dtype)
    keep = ops.DropSeed(p, seed=77) if p > 0 else None
    inv = 1.0 / (1.0 - p)
    _, mean, rstd = ops.ln_modulate_fwd(x, w, shift=mod[:, 3 * D:4 * D], scale=mod[:, 4 * D:5 * D], out_dtype=dtype)
    dx1, dw1, dm1 = torch.empty_like(x), torch.zeros(D, device=DEV), torch.zeros(B, 6 * D, device=DEV)
    ops.ln_modulate_bwd(dy, x, w, mod[:, 4 * D:5 * D], mean, rstd, dx1, add, dw1, dm1[:, 3 * D:4 * D], dm1[:, 4 * D:5 * D])
    da1 = ops.gate_residual_bwd(dx1, a, mod[:, 2 * D:3 * D], dm1[:, 2 * D:3 * D], keep, inv)
    dx2, dw2, dm2 = torch.empty_like(x), torch.zeros(D, device=DEV), torch.zeros(B, 6 * D, device=DEV)
    da2 = ops.ln_modulate_bwd_gate(dy, x, w, mod[:, 4 * D:5 * D], mean, rstd, dx2, add, dw2, dm2[:, 3 * D:4 * D],
                                   dm2[:, 4 * D:5 * D], a, mod[:, 2 * D:3 * D], dm2[:, 2 * D:3 * D], keep, inv)
    assert torch.equal(dx2, dx1) and torch.equal(da2, da1)
    assert torch.equal(dm2[:, 3 * D:5 * D], dm1[:, 3 * D:5 * D])
    assert rel_err(dw2, dw1) < 1e-5                                   # (atomic adds over the batch: order varies)
    assert rel_err(dm2[:, 2 * D:3 * D], dm1[:, 2 * D:3 * D]) < 1e-5   # (rows fold in another order than gate_res_bwd's)
    # block-boundary form: the column sums of da as stored (the producing Linear's bias gradient) ride along
    db1, db3 = torch.full((D,), 0.5, device=DEV), torch.full((D,), 0.5, device=DEV)
    ops.gate_residual_bwd(dx1, a, mod[:, 2 * D:3 * D], dm1[:, 2 * D:3 * D], keep, inv, dbias=db1)
    dx3, dw3, dm3 = torch.empty_like(x), torch.zeros(D, device=DEV), torch.zeros(B, 6 * D, device=DEV)
    da3 = ops.ln_modulate_bwd_gate(dy, x, w, mod[:, 4 * D:5 * D], mean, rstd, dx3, add, dw3, dm3[:, 3 * D:4 * D],
                                   dm3[:, 4 * D:5 * D], a, mod[:, 2 * D:3 * D], dm3[:, 2 * D:3 * D], keep, inv, dbias=db3)
    assert torch.equal(dx3, dx1) and torch.equal(da3, da1) and torch.equal(dm3, dm2) and torch.equal(dw3, dw2)
    assert rel_err(db3, db1) < 1e-5
    assert rel_err(db3 - 0.5, da1.double().sum((0, 1))) < (1e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("d", [32, 64, 96])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rotary(d, dtype):
    ops = _ops()
    torch.manual_seed(d)
    B, L, H = 2, 50, 3
    qkv = torch.randn(B, L, 3, H, d)
    cos, sin = O.rotary_cos_sin(L, d)
    q, k, v = qkv.unbind(2)
    ref = torch.stack([O.apply_rotary(q, cos, sin), O.apply_rotary(k, cos, sin), v], 2)
    c2, s2 = cos[:, :d // 2].contiguous().to(DEV), sin[:, :d // 2].contiguous().to(DEV)
    x = qkv.to(DEV).to(dtype).contiguous()
    ops.rotary_qkv_(x, c2, s2)
    assert rel_err(x, ref) < (1e-6 if dtype == torch.float32 else 5e-3)
    ops.rotary_qkv_(x, c2, s2, conj=True)  # conjugate rotation undoes it
    assert rel_err(x, qkv) < (1e-6 if dtype == torch.float32 else 1e-2)


def test_gelu_silu_embed_epilogue_colsum_transpose():
    ops = _ops()
    torch.manual_seed(5)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 5e-3)):
        u = (torch.randn(1000, 51, device=DEV) * 2).to(dtype)
        ur = u.float().requires_grad_(True)
        ref = F.gelu(ur, approximate="tanh")
        assert rel_err(ops.gelu_fwd(u), ref) < tol
        dg = torch.randn_like(u)
        ref.backward(dg.float())
        assert rel_err(ops.gelu_bwd(u, dg), ur.grad) < max(tol, 2e-6)
    x = torch.randn(64, 128, device=DEV, requires_grad=True)
    y = F.silu(x)
    assert rel_err(ops.silu_fwd(x.detach()), y) < 1e-6
    dy = torch.randn_like(x)
    y.backward(dy)
    assert rel_err(ops.silu_bwd(x.detach(), dy), x.grad) < 1e-6
    sig = torch.rand(9, device=DEV) * 7
    assert_close(ops.timestep_embedding(sig, 256), O.timestep_embedding(sig.cpu()), 1e-5, 2e-6)
    # embedding fwd / scatter-add bwd (small and large vocab)
    for V in (26, 32, 38, 2365):  # (<= 32: per-wave LDS slices; <= 64: LDS atomics; larger: global atomics)
        tab = torch.randn(V, 768, device=DEV)
        idx = torch.randint(0, V, (5, 333), device=DEV)
        assert torch.equal(ops.embedding_fwd(tab, idx), tab[idx])
        dout = torch.randn(5, 333, 768, device=DEV)
        dt = torch.zeros(V, 768, device=DEV)
        ops.embedding_bwd(dout, idx, dt)
        ref = torch.zeros(V, 768, device=DEV).index_add_(0, idx.reshape(-1), dout.reshape(-1, 768))
        assert rel_err(dt, ref) < 1e-5
    # beyond V * M = 2^22 the owner-computes kernel hands over to the scratch path (V <= 32: run twice, bit-identical) and
    # to the LDS- / global-atomic kernels; (2397, 1700) with half the tokens on the mask row is the owner kernel's (run
    # twice, bit-identical)
    for V, M in ((26, 170000), (40, 110000), (300, 15000), (2397, 1700), (2397, 16384)):
        idx = torch.randint(0, V, (M,), device=DEV)
        if V == 2397:
            idx[torch.rand(M, device=DEV) < 0.5] = V - 1
        dout = torch.randn(M, 64, device=DEV)
        ref = torch.zeros(V, 64, device=DEV, dtype=torch.float64).index_add_(0, idx, dout.double())
        got = []
        for _ in range(2):
            dt = torch.zeros(V, 64, device=DEV)
            ops.embedding_bwd(dout, idx, dt)
            got.append(dt)
        assert rel_err(got[0], ref.float()) < 1e-5, V
        if V <= 32 or V * M <= 2 ** 22:
            assert torch.equal(got[0], got[1])
    # logits epilogue (16-byte chunks straddle rows and batch elements; tail chunk; unaligned base -> scalar kernel)
    for dtype in (torch.float32, torch.bfloat16):
        for (B, Ln, V) in ((4, 33, 26), (5, 1, 3), (3, 7, 37)):
            lg = torch.randn(B, Ln, V).to(dtype)
            sg = torch.tensor([0.01, 0.3, 0.7, 5.0, 1.1])[:B]
            idx = torch.randint(0, V, (B, Ln))
            ref = O.logits_epilogue(lg, sg, idx)
            out = ops.logits_epilogue_(lg.to(DEV).clone(), sg.to(DEV), idx.to(DEV))
            assert rel_err(out.float(), ref.float()) < (1e-6 if dtype == torch.float32 else 8e-3)
            assert float(out.gather(-1, idx.to(DEV)[..., None]).abs().max()) == 0.0
            buf = torch.empty(lg.numel() + 8, dtype=dtype, device=DEV)
            mis = buf[1:1 + lg.numel()].view(lg.shape)
            mis.copy_(lg)
            assert torch.equal(ops.logits_epilogue_(mis, sg.to(DEV), idx.to(DEV)), out)
            nos = ops.logits_epilogue_(lg.to(DEV).clone(), sg.to(DEV), idx.to(DEV), scale_by_sigma=False)
            assert torch.equal(nos, lg.to(DEV).scatter(-1, idx.to(DEV)[..., None], 0.0))
        d = ops.logits_epilogue_bwd_(torch.ones(4, 33, 26, device=DEV, dtype=dtype), idx.new_zeros(4, 33).to(DEV))
        assert float(d.float().sum()) == 4 * 33 * 25
    a = torch.randn(777, 130, device=DEV)
    assert rel_err(ops.colsum(a), a.sum(0)) < 1e-5
    assert rel_err(ops.colsum(a.bfloat16()), a.bfloat16().float().sum(0)) < 1e-5
    for Mc, Nc in ((2048, 1024), (515, 300), (8192, 256), (7, 4096)):  # the one-launch form for a few thousand rows
        for dt in (torch.float32, torch.bfloat16):
            t = torch.randn(Mc, Nc + 8, device=DEV).to(dt)[:, 4:4 + Nc]  # (strided rows)
            got = ops.colsum(t)
            assert rel_err(got, t.double().sum(0)) < 1e-5
            assert torch.equal(got, ops.colsum(t)), "fixed summation order"
            acc2 = torch.full((Nc,), 2.0, device=DEV)
            ops.colsum(t, out=acc2, accumulate=True)
            assert rel_err(acc2, 2.0 + t.double().sum(0)) < 1e-5
    acc = torch.ones(130, device=DEV)
    ops.colsum(a, out=acc, accumulate=True)
    assert rel_err(acc, a.sum(0) + 1) < 1e-5
    for dt_in, dt_out in ((torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)):
        a = torch.randn(130, 77, device=DEV).to(dt_in)
        assert torch.equal(ops.transpose(a, dt_out), a.T.to(dt_out).contiguous())
    assert torch.equal(ops.cast(a.float(), torch.bfloat16), a.float().bfloat16())


def test_hashed_dropout_is_consistent_between_forward_and_backward():
    """counter-based dropout (seed instead of a stored mask): keep rate = 1 - p, the backward regenerates the forward's
    mask exactly, different seeds give different masks, and the result equals the explicit-mask path."""
    ops = _ops()
    B, L, D, p = 3, 70, 768, 0.1
    inv = 1.0 / (1.0 - p)
    ones = torch.ones(B, L, D, device=DEV)
    gate = torch.ones(B, D, device=DEV)
    ds = ops.DropSeed(p, seed=12345)
    mask = ops.gate_residual_fwd(torch.zeros_like(ones), ones, gate, ds, inv) > 0
    frac = float(mask.float().mean())
    assert abs(frac - (1 - p)) < 0.005, frac
    assert abs(float(mask.float().mean(dim=(0, 1)).min()) - (1 - p)) < 0.15  # no dead columns
    mask2 = ops.gate_residual_fwd(torch.zeros_like(ones), ones, gate, ops.DropSeed(p, seed=12346), inv) > 0
    assert 0.1 < float((mask ^ mask2).float().mean()) < 0.3
    torch.manual_seed(0)
    res, a = torch.randn(B, L, D, device=DEV), torch.randn(B, L, D, device=DEV).bfloat16()
    g = torch.randn(B, 6 * D, device=DEV).bfloat16()[:, 2 * D:3 * D]
    dout = torch.randn(B, L, D, device=DEV)
    keep = mask.to(torch.uint8)
    assert torch.equal(ops.gate_residual_fwd(res, a, g, ds, inv), ops.gate_residual_fwd(res, a, g, keep, inv))
    dg1, dg2 = torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    da1 = ops.gate_residual_bwd(dout, a, g, dg1, ds, inv)
    da2 = ops.gate_residual_bwd(dout, a, g, dg2, keep, inv)
    assert torch.equal(da1, da2) and torch.equal(dg1, dg2)


def _ref_attn(q, k, v, mask):
    return O.softmax_attention(q, k, v, mask)


@pytest.mark.parametrize("d", [32, 64, 96])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Lq,Lk,masked", [(128, 128, False), (70, 70, False), (100, 45, True), (4, 4, False), (200, 130, True),
                                          (16, 9, True), (3, 16, True), (1, 2, False), (17, 16, False)])
def test_attention(d, dtype, Lq, Lk, masked):
    ops = _ops()
    torch.manual_seed(d + Lq)
    B, H = 2, 3
    q = torch.randn(B, Lq, H, d)
    k = torch.randn(B, Lk, H, d)
    v = torch.randn(B, Lk, H, d)
    if dtype == torch.bfloat16:
        q, k, v = (t.bfloat16().float() for t in (q, k, v))
    mask = None
    if masked:
        mask = torch.rand(B, Lk) > 0.3
        mask[:, 0] = True
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attn(qr, kr, vr, mask)
    dout = torch.randn(B, Lq, H, d)
    if dtype == torch.bfloat16:
        dout = dout.bfloat16().float()
    ref.backward(dout)
    qd, kd, vd = (t.to(DEV).to(dtype) for t in (q, k, v))
    md = None if mask is None else mask.to(DEV)
    o, lse = ops.attn_fwd(qd, kd, vd, md)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(o, ref) < tol, "o"
    s = torch.einsum("blhd,bmhd->bhlm", q, k) * d ** -0.5
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    # bf16: the row sum is taken (on the matrix pipe) over the bf16-rounded probabilities that P V multiplies with
    lse_tol = 1e-4 if dtype == torch.float32 else 2e-3
    assert_close(lse, torch.logsumexp(s, -1), lse_tol, lse_tol, "lse")
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attn_bwd(qd, kd, vd, o, dout.to(DEV).to(dtype), lse, dq, dk, dv, md)
    tolb = 5e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(dq, qr.grad) < tolb, "dq"
    assert rel_err(dk, kr.grad) < tolb, "dk"
    assert rel_err(dv, vr.grad) < tolb, "dv"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_over_the_batch_axis(dtype):
    """the dual-track model's self-attention as written (quirk Q1, transformers_protlig_cross.py:305: batch_first MHA on
    [L, B, D]): 'sequence' = the batch index (4 here), 'batch' = the positions — transposed strided views, many tiny
    problems (the one-thread-per-row kernels for d = 32), outputs written through the same transposed views."""
    ops = _ops()
    torch.manual_seed(5)
    Bt, L, H, d = 4, 37, 5, 32
    q, k, v, dout = (torch.randn(Bt, L, H, d) for _ in range(4))
    if dtype == torch.bfloat16:
        q, k, v, dout = (t.bfloat16().float() for t in (q, k, v, dout))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attn(qr.transpose(0, 1), kr.transpose(0, 1), vr.transpose(0, 1), None)  # [L, Bt, H, d]: attends over Bt
    ref.backward(dout.transpose(0, 1))
    qd, kd, vd, dd = (t.to(DEV).to(dtype) for t in (q, k, v, dout))
    o = torch.empty_like(qd)
    _, lse = ops.attn_fwd(qd.transpose(0, 1), kd.transpose(0, 1), vd.transpose(0, 1), out=o.transpose(0, 1))
    tol, tolb = (2e-5, 5e-5) if dtype == torch.float32 else (1e-2, 2e-2)
    assert rel_err(o.transpose(0, 1), ref) < tol
    s = torch.einsum("blhd,bmhd->bhlm", q.transpose(0, 1), k.transpose(0, 1)) * d ** -0.5
    assert_close(lse, torch.logsumexp(s, -1), 2e-3, 2e-3, "lse")
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attn_bwd(qd.transpose(0, 1), kd.transpose(0, 1), vd.transpose(0, 1), o.transpose(0, 1), dd.transpose(0, 1), lse,
                 dq.transpose(0, 1), dk.transpose(0, 1), dv.transpose(0, 1))
    assert rel_err(dq, qr.grad) < tolb and rel_err(dk, kr.grad) < tolb and rel_err(dv, vr.grad) < tolb
    if dtype == torch.bfloat16:
        # o / dout rows that are only 8-byte aligned (row stride % 8 == 4 elements): allowed by the ABI, not by the
        # one-thread-per-row kernels' 16-byte accesses -> the tile kernels must take the call, same results
        def odd(t):
            buf = torch.zeros(Bt, L, H * d + 4, device=DEV, dtype=dtype)
            view = buf[:, :, :H * d].view(Bt, L, H, d)
            view.copy_(t)
            return view
        o2, dd2 = odd(torch.zeros_like(qd)), odd(dd)
        assert o2.stride(1) % 8 == 4
        _, lse2 = ops.attn_fwd(qd.transpose(0, 1), kd.transpose(0, 1), vd.transpose(0, 1), out=o2.transpose(0, 1))
        assert rel_err(o2.transpose(0, 1), ref) < tol
        dq2, dk2, dv2 = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        ops.attn_bwd(qd.transpose(0, 1), kd.transpose(0, 1), vd.transpose(0, 1), o2.transpose(0, 1), dd2.transpose(0, 1),
                     lse2, dq2.transpose(0, 1), dk2.transpose(0, 1), dv2.transpose(0, 1))
        assert rel_err(dq2, qr.grad) < tolb and rel_err(dk2, kr.grad) < tolb and rel_err(dv2, vr.grad) < tolb


def test_attention_packed_qkv_and_spike():
    """packed [B, L, 3, H, d] views (the DDiTBlock layout) and an input that forces the online-softmax rescale
    (one key row spiked against one query so the running max jumps in a later tile)."""
    ops = _ops()
    torch.manual_seed(11)
    B, L, H, d = 2, 256, 4, 64
    qkv = torch.randn(B, L, 3, H, d)
    qkv[0, 200, 1, 2] = qkv[0, 7, 0, 2] * 6  # key 200 (4th tile) aligned with query 7
    qkv = qkv.bfloat16()
    x = qkv.to(DEV)
    o, lse = ops.attn_fwd(x[:, :, 0], x[:, :, 1], x[:, :, 2])
    ref = _ref_attn(*(qkv.float().unbind(2)), None)
    assert rel_err(o, ref) < 1e-2
    assert float((o[0, 7, 2].float().cpu() - ref[0, 7, 2]).abs().max()) < 3e-2
    dqkv = torch.empty_like(x)
    dout = torch.randn(B, L, H, d).bfloat16()
    ops.attn_bwd(x[:, :, 0], x[:, :, 1], x[:, :, 2], o, dout.to(DEV), lse, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
    qr = qkv.float().requires_grad_(True)
    _ref_attn(*qr.unbind(2), None).backward(dout.float())
    assert rel_err(dqkv, qr.grad) < 2e-2


@pytest.mark.parametrize("dtype,B,L,H,d", [(torch.bfloat16, 2, 256, 4, 64), (torch.bfloat16, 1, 200, 2, 32),
                                           (torch.float32, 1, 96, 2, 96), (torch.bfloat16, 1, 128, 1, 128)])
def test_attention_backward_fused_rotary_gradient(dtype, B, L, H, d):
    """attn_bwd(rotary=(cos, sin)) == attn_bwd followed by rotary_qkv_(conj=True) (rotary.py:42-51's gradient)."""
    ops = _ops()
    torch.manual_seed(d + L)
    x = torch.randn(B, L, 3, H, d, device=DEV).to(dtype)
    dout = torch.randn(B, L, H, d, device=DEV).to(dtype)
    ang = torch.rand(L, d // 2, device=DEV) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    o, lse = ops.attn_fwd(x[:, :, 0], x[:, :, 1], x[:, :, 2])
    ref = torch.empty_like(x)
    ops.attn_bwd(x[:, :, 0], x[:, :, 1], x[:, :, 2], o, dout, lse, ref[:, :, 0], ref[:, :, 1], ref[:, :, 2])
    ops.rotary_qkv_(ref, cos, sin, conj=True)
    got = torch.empty_like(x)
    ops.attn_bwd(x[:, :, 0], x[:, :, 1], x[:, :, 2], o, dout, lse, got[:, :, 0], got[:, :, 1], got[:, :, 2],
                 rotary=(cos, sin))
    assert torch.equal(got[:, :, 2], ref[:, :, 2])
    # same fp32 arithmetic on the same rounded inputs; fma contraction may differ by an ulp of the output type
    assert rel_err(got.float(), ref.float()) < (1e-6 if dtype == torch.float32 else 4e-3)


def test_optimizer_kernels():
    ops = _ops()
    torch.manual_seed(7)
    n = 4096 * 37 + 8
    p = torch.randn(n)
    g = torch.randn(n) * 3
    m, v, ema = torch.zeros(n), torch.zeros(n), p.clone()
    pd, gd, md, vd, ed = (t.to(DEV).clone() for t in (p, g, m, v, ema))
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    lr, b1, b2, eps, wd, decay = 3e-4, 0.9, 0.999, 1e-8, 0.01, 0.9999
    for step in (1, 2, 3):
        out = ops.sumsq_clip(gd, 1.0, 1.0)
        gn = float(torch.sqrt((g.double() ** 2).sum()))
        assert abs(float(out[0]) ** 0.5 - gn) / gn < 1e-5
        coef = O.clip_coef(gn, 1.0)
        assert abs(float(out[1]) - coef) / coef < 1e-5
        O.adamw_step(p, g, m, v, step, lr, b1, b2, eps, wd, coef)
        O.ema_update(ema, p, O.ema_decay(step, decay))
        ops.adamw_ema_step(pd, gd, md, vd, ed, pb, out[1:], 1.0, lr, b1, b2, eps, wd, step, O.ema_decay(step, decay))
        assert_close(pd, p, 1e-5, 1e-6, f"p step {step}")
        assert_close(md, m, 1e-5, 1e-7)
        assert_close(vd, v, 1e-5, 1e-9)
        assert_close(ed, ema, 1e-5, 1e-6)
        assert torch.equal(pb, pd.bfloat16())


def test_baseline_size_properties_attention_and_gemm():
    """Size-independent properties at BASELINE sizes (where the CPU oracle would take minutes): attention at the cfg-2 /
    cfg-5 shapes with V = 1 returns exactly 1 and is exactly linear in V; the big-tile GEMMs at the cfg-2 MLP shape are
    exactly linear under power-of-two scaling, agree between the NT form and the NN form on a transposed copy of the
    weight, and the weight-gradient (TN) form equals the NT product of the transposed operands."""
    ops = _ops()
    torch.manual_seed(7)
    for B, H, L, d in ((64, 12, 512, 64), (32, 12, 1024, 64)):
        qkv = torch.randn(B, L, 3, H, d, device=DEV, dtype=torch.bfloat16)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        o, lse = ops.attn_fwd(q, k, v)
        assert bool(torch.isfinite(lse).all())
        qkv1 = qkv.clone()
        qkv1[:, :, 2] = 1.0
        o1, _ = ops.attn_fwd(qkv1[:, :, 0], qkv1[:, :, 1], qkv1[:, :, 2])
        assert torch.equal(o1, torch.ones_like(o1)), "softmax rows must sum to one"
        qkv2 = qkv.clone()
        qkv2[:, :, 2] *= 2
        o2, _ = ops.attn_fwd(qkv2[:, :, 0], qkv2[:, :, 1], qkv2[:, :, 2])
        assert torch.equal(o2.float(), o.float() * 2), "attention is linear in V"
        # every output is a convex combination of the values of its head
        vmax = v.float().amax(dim=1, keepdim=True)
        vmin = v.float().amin(dim=1, keepdim=True)
        assert bool(((o.float() <= vmax + 1e-2) & (o.float() >= vmin - 1e-2)).all())
    M, D, F = 64 * 512, 768, 3072
    x = torch.randn(M, D, device=DEV).bfloat16()
    W = (torch.randn(F, D, device=DEV) * 0.05).bfloat16()
    y = ops.gemm_nt(x, W)
    assert torch.equal(ops.gemm_nt(x, (W.float() * 4).bfloat16()).float(), y.float() * 4)
    Wt = W.t().contiguous()                       # [D, F]: the NN form multiplies by B as stored
    assert rel_err(ops.gemm_nn(x, Wt), y) < 1e-6   # same products, same fp32 accumulation order per K tile
    g = torch.randn(M, F, device=DEV).bfloat16()
    dW = ops.gemm_tn(g, x, out_dtype=torch.float32)            # [F, D] = g^T x
    ref = ops.gemm_nt(g.t().contiguous(), x.t().contiguous(), out_dtype=torch.float32)
    assert rel_err(dW, ref) < 1e-5                # (split-K changes the fp32 summation order)
