"""Host-side cost of a cfg-3 train step by phase, measured with tiny inputs so the GPU never holds the host back."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import graph_lib, noise_lib, ops
from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
from protein_lig_sedd_amd.train import CrossAttTrainer
ns = types.SimpleNamespace
dev = torch.device("cuda", 0)
for name in ("fp32", "bf16"):
    dt = dict(fp32=torch.float32, bf16=torch.bfloat16)[name]
    cfg = ns(graph=ns(type="absorb"), model=ns(hidden_size=1024, cond_dim=128, n_heads=32, n_blocks=12, dropout=0.1, scale_by_sigma=True))
    model = ProteinLigandDiffusionModel(cfg, compute_dtype=dt).to(dev)
    tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(dev), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                         dict(lr=1e-4, warmup=1000, grad_clip=1.0), task="joint", world_size=1)
    b = dict(prot_tokens=torch.randint(0, 36, (1, 64), device=dev), lig_tokens=torch.randint(0, 2364, (1, 16), device=dev))
    for _ in range(3): tr.train_step(b)
    torch.cuda.synchronize()
    acc = [0.0, 0.0, 0.0]
    n = 10
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.flat.g.zero_(); model.train()
        out = tr.calc_loss(b, "joint"); loss = out[0].mean()
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        ops.sumsq_clip(tr.flat.g, 1.0, 1.0, tr._ws, tr._norm)
        tr.flat.refresh_compute_weights()
        t3 = time.perf_counter()
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
    print(f"{name}: forward+loss {acc[0]/n*1e3:.2f} ms, backward {acc[1]/n*1e3:.2f} ms, clip+refresh {acc[2]/n*1e3:.2f} ms (host time, enqueue only)")
    del tr, model
