"""HBM-bound kernels at BASELINE cfg-5 shard size (512 x 1024 rows, V = 26) and cfg-3 ligand vocab (V = 2365):
achieved GB/s = algorithmic bytes / HIP-event time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import ops, graph_lib, _lib as L

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def report(name, nbytes, ms):
    print(f"{name:44s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.0f} GB/s  {nbytes/ms/1e6/8000*100:5.1f}% of 8 TB/s", flush=True)

dev = "cuda"
for (B, Ln, V, dt) in ((512, 1024, 26, torch.bfloat16), (512, 1024, 26, torch.float32), (64, 128, 2365, torch.float32)):
    g = graph_lib.Absorbing(V - 1)
    M = B * Ln
    e = 2 if dt == torch.bfloat16 else 4
    x = torch.randint(0, V, (B, Ln), device=dev); x[torch.rand(B, Ln, device=dev) < 0.5] = V - 1
    x0 = torch.randint(0, V - 1, (B, Ln), device=dev)
    logits = torch.randn(B, Ln, V, device=dev).to(dt)
    u = torch.rand(B, Ln, V, device=dev)
    sig = torch.rand(B, device=dev) + 0.1
    tag = f"[{B}x{Ln}, V={V}, {'bf16' if e == 2 else 'fp32'}]"
    for nm, st in (("euler", L.STEP_EULER), ("analytic", L.STEP_ANALYTIC)):
        ms = timeit(lambda: ops.predictor_step(g, st, logits, x, sig, u))
        report(f"predictor_step {nm} {tag}", M * (V * e + V * 4 + 16), ms)
    ud = u[..., :-1].contiguous()
    ms = timeit(lambda: ops.predictor_step(g, L.STEP_DENOISE, logits, x, sig, ud))
    report(f"predictor_step denoise {tag}", M * (V * e + (V - 1) * 4 + 16), ms)
    ms = timeit(lambda: ops.score_entropy_fwd(g, logits, sig, x, x0))
    report(f"score_entropy fwd {tag}", M * (V * e + 20), ms)
    ms = timeit(lambda: ops.score_entropy_fwd(g, logits, sig, x, x0, gw=sig))
    report(f"score_entropy fwd+bwd {tag}", M * (2 * V * e + 20), ms)
    u1 = u[..., 0].contiguous()
    ms = timeit(lambda: ops.sample_transition(g, x0, sig, u1))
    report(f"sample_transition {tag}", M * 20, ms)
    ms = timeit(lambda: ops.gumbel_argmax(logits, u))
    report(f"gumbel_argmax {tag}", M * (V * e + V * 4 + 8), ms)
    ms = timeit(lambda: ops.logits_epilogue_(logits, sig, x))
    report(f"logits_epilogue {tag}", M * (2 * V * e + 8), ms)
# model elementwise at cfg-2 size
B, Ln, D = 64, 512, 768
M = B * Ln
x = torch.randn(B, Ln, D, device=dev); w = torch.ones(D, device=dev)
mod = torch.randn(B, 6 * D, device=dev).bfloat16()
ms = timeit(lambda: ops.ln_modulate_fwd(x, w, shift=mod[:, :D], scale=mod[:, D:2*D], out_dtype=torch.bfloat16))
report("ln_modulate_fwd [64x512x768] bf16 out", M * D * 6, ms)
a = torch.randn(B, Ln, D, device=dev).bfloat16()
ms = timeit(lambda: ops.gate_residual_fwd(x, a, mod[:, 2*D:3*D]))
report("gate_residual_fwd", M * D * 10, ms)
qkv = torch.randn(B, Ln, 3, 12, 64, device=dev).bfloat16()
cs = torch.rand(Ln, 32, device=dev); sn = torch.rand(Ln, 32, device=dev)
ms = timeit(lambda: ops.rotary_qkv_(qkv, cs, sn))
report("rotary_qkv (q,k in place)", 2 * M * 2 * D * 2, ms)
big = torch.randn(M, 3072, device=dev).bfloat16()
ms = timeit(lambda: ops.transpose(big)); report("transpose [32768,3072] bf16", 2 * big.numel() * 2, ms)
ms = timeit(lambda: ops.colsum(big)); report("colsum [32768,3072] bf16", big.numel() * 2, ms)
ms = timeit(lambda: ops.gelu_bwd(big, big)); report("gelu_bwd [32768,3072] bf16", 3 * big.numel() * 2, ms)
# backward elementwise (cfg-2 size)
dyb = torch.randn(B, Ln, D, device=dev).bfloat16()
_, mean, rstd = ops.ln_modulate_fwd(x, w, shift=mod[:, :D], scale=mod[:, D:2*D], out_dtype=torch.bfloat16)
dxo, dxa = torch.empty_like(x), torch.randn_like(x)
dw = torch.zeros(D, device=dev); dmod = torch.empty(B, 6 * D, device=dev)
ms = timeit(lambda: ops.ln_modulate_bwd(dyb, x, w, mod[:, D:2*D], mean, rstd, dxo, dxa, dw, dmod[:, :D], dmod[:, D:2*D]))
report("ln_modulate_bwd (+reduce) dy bf16, dx += add", M * D * 14, ms)
ms = timeit(lambda: ops.gate_residual_bwd(dxa, a, mod[:, 2*D:3*D], dmod[:, 2*D:3*D], None, 1.0))
report("gate_residual_bwd (+reduce)", M * D * 8, ms)
