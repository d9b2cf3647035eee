"""sedd_ln_modulate_bwd_noalloc (the one entry point that allocates nothing: its [3, nchunk, D] partial-sum workspace is
a file-scope device array of the library) against the existing pair sedd_ln_modulate_bwd + sedd_ln_modulate_bwd_reduce
with a caller workspace, through the C ABI: two calls in a row, the second with a smaller input (the shared array keeps
the first call's contents), bit-exact — both run the same deterministic kernels — plus the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, L, D, dy_dtype, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, D, generator=g).to(DEV)
    dy = torch.randn(B, L, D, generator=g).to(DEV).to(dy_dtype)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    mod = (0.1 * torch.randn(B, 2 * D, generator=g)).to(DEV)
    add = torch.randn(B, L, D, generator=g).to(DEV)
    return x, dy, w, mod, add


def _run(noalloc, x, dy, w, mod, add):
    from protein_lig_sedd_amd import _lib, ops
    L_ = _lib.lib()
    B, L, D = x.shape
    M = B * L
    y, mean, rstd = ops.ln_modulate_fwd(x, w, shift=mod[:, :D], scale=mod[:, D:], out_dtype=dy.dtype)
    dx = torch.empty_like(x)
    dw = torch.full((D,), 0.25, device=DEV)          # accumulated into (+=)
    dmod = torch.zeros(B, 2 * D, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    scale = mod[:, D:]
    args = (dy.data_ptr(), ops._dt(dy), x.data_ptr(), w.data_ptr(), scale.data_ptr(), ops._dt(mod), mod.stride(0),
            mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), add.data_ptr())
    if noalloc:
        _lib.check(L_.sedd_ln_modulate_bwd_noalloc(*args, dw.data_ptr(), dmod[:, :D].data_ptr(), dmod[:, D:].data_ptr(),
                                                   dmod.stride(0), M, L, D, st), "sedd_ln_modulate_bwd_noalloc")
    else:
        ws = torch.empty(3 * L_.sedd_ln_bwd_chunks(M, L) * D, device=DEV)
        _lib.check(L_.sedd_ln_modulate_bwd(*args, ws.data_ptr(), M, L, D, st), "sedd_ln_modulate_bwd")
        _lib.check(L_.sedd_ln_modulate_bwd_reduce(ws.data_ptr(), dw.data_ptr(), dmod[:, :D].data_ptr(),
                                                  dmod[:, D:].data_ptr(), dmod.stride(0), M, L, D, st),
                   "sedd_ln_modulate_bwd_reduce")
    torch.cuda.synchronize()
    return dx, dw, dmod


def test_noalloc_matches_the_workspace_entry_points_twice():
    # first call: cfg-2 sized rows (B 8 x L 512, D 768, bf16 dy: 64 chunks); second: smaller, fp32 dy, ragged L, D 1024
    for B, L, D, dt, seed in ((8, 512, 768, torch.bfloat16, 1), (3, 100, 1024, torch.float32, 2)):
        case = _case(B, L, D, dt, seed)
        ref = _run(False, *case)
        mem0 = torch.cuda.memory_allocated()
        got = _run(True, *case)
        for name, a, b in zip(("dx", "dw", "dshift|dscale"), got, ref):
            assert torch.equal(a, b), f"B{B} L{L} D{D} {name}: max |diff| {float((a - b).abs().max()):.3e}"
        assert torch.isfinite(got[0]).all() and float(got[2].abs().max()) > 0
        del got
        assert torch.cuda.memory_allocated() <= mem0 + 4 * (B * L * D + D + 2 * B * D) + (1 << 16)  # outputs only


def test_noalloc_rejects_sizes_beyond_its_scratch():
    from protein_lig_sedd_amd import _lib
    L_ = _lib.lib()
    D, L = 1024, 64
    B = 4 * 1024 * 1024 // (3 * D) + 1              # from 16384 rows up chunks hold 64 rows: one per batch element
    M = B * L
    assert 3 * L_.sedd_ln_bwd_chunks(M, L) * D > 4 * 1024 * 1024
    p = torch.empty(16, device=DEV).data_ptr()
    rc = L_.sedd_ln_modulate_bwd_noalloc(p, 0, p, p, p, 0, 2 * D, p, p, p, None, p, p, p, 2 * D, M, L, D, None)
    assert rc == -2 and b"static scratch" in L_.sedd_last_error()
