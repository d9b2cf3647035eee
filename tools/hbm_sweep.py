"""M-sweep of the HBM-bound loss / sampling kernels (V = 26, bf16 log-scores) from the cfg-5 shard (0.5 M rows) to 8 M rows,
beside a plain device copy moving the SAME number of bytes (read + write) — the yardstick for what a launch of that size
can reach on this chip: a 20-30 us launch spends a fixed few microseconds ramping up and draining."""
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
    return 1e3 * s.elapsed_time(e) / n

dev, V = "cuda", 26
g = graph_lib.Absorbing(V - 1)
print(f"{'rows':>9s} | {'kernel':28s} {'us':>8s} {'GB/s':>7s} {'of 8TB/s':>8s} | copy of the same bytes: us, GB/s", flush=True)
for B in (512, 1024, 2048, 4096, 8192):
    Ln = 1024
    M = B * Ln
    x = torch.randint(0, V, (B, Ln), device=dev); x[torch.rand(B, Ln, device=dev) < 0.5] = V - 1
    x0 = torch.randint(0, V - 1, (B, Ln), device=dev)
    logits = torch.randn(B, Ln, V, device=dev).bfloat16()
    u = torch.rand(B, Ln, V, device=dev)
    sig = torch.rand(B, device=dev) + 0.1
    cases = (("score_entropy fwd+bwd", M * (2 * V * 2 + 20), lambda: ops.score_entropy_fwd(g, logits, sig, x, x0, gw=sig)),
             ("predictor analytic", M * (V * 2 + V * 4 + 16), lambda: ops.predictor_step(g, L.STEP_ANALYTIC, logits, x, sig, u)),
             ("predictor analytic philox", M * (V * 2 + 16), lambda: ops.predictor_step_philox(g, L.STEP_ANALYTIC, logits, x, sig, ops.PhiloxNoise(1, 7))),
             ("gumbel_argmax", M * (V * 2 + V * 4 + 8), lambda: ops.gumbel_argmax(logits, u)))
    for name, nbytes, fn in cases:
        us = timeit(fn)
        half = nbytes // 2 // 16 * 16
        src, dst = torch.empty(half, dtype=torch.uint8, device=dev), torch.empty(half, dtype=torch.uint8, device=dev)
        cus = timeit(lambda: dst.copy_(src))
        print(f"{M:9d} | {name:28s} {us:8.1f} {nbytes / us / 1e3:7.0f} {nbytes / us / 1e3 / 8000:8.2f} | {cus:7.1f} {2 * half / cus / 1e3:7.0f}", flush=True)
        del src, dst
