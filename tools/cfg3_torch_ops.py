"""Which torch operators (not this package's kernels) still launch inside a cfg-3 train step, and from where: a torch.profiler
pass over a few steps, grouped by operator and by Python stack.  The step is bound by the host's enqueue rate at batch 4, so
every stray copy / fill / add is worth its launch."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from protein_lig_sedd_amd import graph_lib, noise_lib
from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
from protein_lig_sedd_amd.train import CrossAttTrainer
ns = types.SimpleNamespace
dev = torch.device("cuda", 0)
D, H, C, nb, Lp, Ll, B = 1024, 32, 128, 12, 512, 128, 4
cfg = ns(graph=ns(type="absorb"), model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=nb, dropout=0.1, scale_by_sigma=True))
torch.manual_seed(0)
model = ProteinLigandDiffusionModel(cfg, compute_dtype=torch.bfloat16).to(dev)
tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(dev), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                     dict(lr=1e-4, warmup=1000, grad_clip=1.0), task="joint", ema_decay=0.9999, world_size=1)
bs = [dict(prot_tokens=torch.randint(0, 36, (B, Lp), device=dev), lig_tokens=torch.randint(0, 2364, (B, Ll), device=dev)) for _ in range(4)]
for i in range(4): tr.train_step(bs[i % 4])
torch.cuda.synchronize()
n = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for i in range(n): tr.train_step(bs[i % 4])
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=8)
rows = [e for e in ka if e.key.startswith("aten::") and e.key in (
    "aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::fill_", "aten::zero_", "aten::zeros",
    "aten::ones", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::bernoulli_", "aten::rand", "aten::uniform_",
    "aten::sum", "aten::mean", "aten::div", "aten::sub", "aten::neg", "aten::exp", "aten::log1p", "aten::cat", "aten::zeros_like")]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
    st = [s for s in e.stack if "protein_lig_sedd_amd" in s or "tools/" in s][:3]
    print(f"{e.count / n:7.1f}/step  {e.key:18s} {' <- '.join(s.split('protein_lig_sedd_amd/')[-1] for s in st)}")
