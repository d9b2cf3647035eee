"""Run-to-run reproducibility of the cross model's joint backward at BASELINE configs[2] dims (2 blocks per track): the
same forward + backward twice from the same state; lists the gradient tensors that differ bitwise."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle.weights import seeded_state_dict
from test_oracle_vs_golden import cross_shapes
from protein_lig_sedd_amd import graph_lib
from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
DEV = "cuda"
ns = types.SimpleNamespace
D, H, C, nb, B, Lp, Ll = 1024, 32, 128, 2, 4, 512, 128
cfg = ns(model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks_prot=nb, n_blocks_lig=nb, n_blocks=nb, dropout=0.0,
                  scale_by_sigma=True, seq_axis_attention=False), tokens_prot=36, tokens_lig=2364)
def run():
    from test_gpu_cross_model import _cfg, load_sd
    model = ProteinLigandDiffusionModel(_cfg(D, H, C, nb)).to(DEV)
    load_sd(model, seeded_state_dict(cross_shapes(D, C, nb), seed=33, std=0.03))
    gen = torch.Generator().manual_seed(33)
    pt = torch.randint(0, 37, (B, Lp), generator=gen).to(DEV); p0 = torch.randint(0, 33, (B, Lp), generator=gen).to(DEV)
    lt = torch.randint(0, 2365, (B, Ll), generator=gen).to(DEV); l0 = torch.randint(0, 2362, (B, Ll), generator=gen).to(DEV)
    sg = (torch.rand(B, generator=gen) * 2 + 0.1).to(DEV)
    model.train()
    lp, ll = model(protein_indices=pt, ligand_indices=lt, timesteps=sg, mode="joint")
    gp, gl = graph_lib.Absorbing(36), graph_lib.Absorbing(2364)
    loss = (gp.score_entropy(lp, sg[:, None], pt, p0).mean() + gl.score_entropy(ll, sg[:, None], lt, l0).mean()) / 2
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
la, ga = run()
lb, gb = run()
bad = [(k, float((ga[k] - gb[k]).abs().max())) for k in ga if not torch.equal(ga[k], gb[k])]
print("loss", la, lb, f"{len(bad)} of {len(ga)} gradient tensors differ")
for k, d in bad[:40]: print("  ", k, f"{d:.3e}")
