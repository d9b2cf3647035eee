"""BASELINE configs[3] (cfg 4: D 1024 x 16 blocks, 16 heads, L 1024, per-GPU batch 16, bf16) train step time on one GPU."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from protein_lig_sedd_amd import graph_lib, noise_lib
from protein_lig_sedd_amd.model.transformer import _SEDD
from protein_lig_sedd_amd.train import SEDDTrainer
ns = types.SimpleNamespace
dev = torch.device("cuda", 0)
V = int(os.environ.get("CFG4_TOKENS", "25"))
cfg = ns(tokens=V, graph=ns(type="absorb"), model=ns(hidden_size=1024, cond_dim=128, n_heads=16, n_blocks=16, dropout=0.1, scale_by_sigma=True))
torch.manual_seed(0)
model = _SEDD(cfg, compute_dtype=torch.bfloat16).to(dev)
bench.randomize(model, 1)
graph, noise = graph_lib.Absorbing(V), (noise_lib.CosineNoise(sigma_min=1e-3, sigma_max=1.0) if os.environ.get("CFG4_NOISE") == "cosine" else noise_lib.LogLinearNoise(eps=1e-3)).to(dev)
tr = SEDDTrainer(model, noise, graph, dict(lr=3e-4, warmup=100, grad_clip=1.0), ema_decay=0.9999, world_size=1)
B, L = 16, 1024
xs = [torch.randint(0, V, (B, L), device=dev) for _ in range(4)]
for i in range(3): tr.train_step(xs[i % 4])
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.perf_counter()
for i in range(n): loss = tr.train_step(xs[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
P = sum(p.numel() for p in model.parameters())
fl = 6.0 * B * L * (16 * 12 * 1024 * 1024) + 12.0 * 16 * B * 16 * L * L * 64  # GEMM fwd+bwd + attention fwd+bwd (3.5x)... rough
print(f"cfg4 (tokens {V}): {dt*1e3:.2f} ms/step, {B*L/dt:.0f} tokens/s, params {P/1e6:.1f} M, loss {float(loss):.4f}, model-flops rate ~{fl/dt/1e12:.0f} TF/s")

if os.environ.get("CFG4_SAMPLE"):
    from protein_lig_sedd_amd import sampling
    model.eval()
    steps = 25
    for pred in ("euler", "analytic"):
        fn = sampling.get_pc_sampler(graph, noise, (B, L), pred, steps, denoise=True, eps=1e-5, device=dev)
        fn(model); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(model); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"  sampling {pred} {steps} steps + denoise, batch {B} x L {L}: {dt*1e3:.1f} ms ({dt/(steps+1)*1e3:.2f} ms/step), tokens in [{int(out.min())}, {int(out.max())}]")
