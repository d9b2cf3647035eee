"""cProfile of the host side of the cfg-3 train step (bf16): where the Python time per launch goes."""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import graph_lib, noise_lib
from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
from protein_lig_sedd_amd.train import CrossAttTrainer
ns = types.SimpleNamespace
dev = torch.device("cuda", 0)
cfg = ns(graph=ns(type="absorb"), model=ns(hidden_size=1024, cond_dim=128, n_heads=32, n_blocks=12, dropout=0.1, scale_by_sigma=True))
model = ProteinLigandDiffusionModel(cfg, compute_dtype=torch.bfloat16).to(dev)
tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(dev), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                     dict(lr=1e-4, warmup=1000, grad_clip=1.0), task="joint", ema_decay=0.9999, world_size=1)
b = dict(prot_tokens=torch.randint(0, 36, (4, 512), device=dev), lig_tokens=torch.randint(0, 2364, (4, 128), device=dev))
for _ in range(3): tr.train_step(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): tr.train_step(b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
