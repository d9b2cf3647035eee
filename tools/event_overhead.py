# measure the step time with and without the per-launch HIP events (same process, same model)
import os, sys, time, types
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
import torch
from protein_lig_sedd_amd import ops, graph_lib, noise_lib
from protein_lig_sedd_amd.model.transformer import _SEDD
from protein_lig_sedd_amd.train import SEDDTrainer
ns = types.SimpleNamespace
cfg = ns(tokens=25, graph=ns(type="absorb"), model=ns(hidden_size=768, cond_dim=128, n_heads=12, n_blocks=12, dropout=0.1, scale_by_sigma=True, length=512))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = _SEDD(cfg, compute_dtype=torch.bfloat16).to(dev)
tr = SEDDTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(dev), graph_lib.Absorbing(25), dict(lr=3e-4, warmup=2500, grad_clip=1.0), world_size=1)
xs = [torch.randint(0, 25, (64, 512), device=dev) for _ in range(4)]
def run(n, prof):
    ops.PROF = ops.KernelProf() if prof else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): tr.train_step(xs[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    ops.PROF = None
    return dt
run(3, False)
for rep in range(3):
    print(f"events on: {run(10, True):.2f} ms/step   events off: {run(10, False):.2f} ms/step", flush=True)
