"""Run-to-run reproducibility of the cfg-2 train step: two trainers built from the same seed take the same three steps
(same batches, same perturbation noise, same dropout seeds); reports which parameter tensors differ bitwise and by how much."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from protein_lig_sedd_amd import graph_lib, noise_lib, ops
from protein_lig_sedd_amd.model.transformer import _SEDD
from protein_lig_sedd_amd.train import SEDDTrainer

dev = torch.device("cuda", 0)
B, L = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (64, 512)

def run():
    torch.manual_seed(1234)
    ops.DropSeed._counter = 0
    cfg = bench.small_config()
    model = _SEDD(cfg, compute_dtype=torch.bfloat16).to(dev)
    bench.randomize(model, 1234)
    graph, noise = graph_lib.Absorbing(cfg.tokens), noise_lib.LogLinearNoise(eps=1e-3).to(dev)
    tr = SEDDTrainer(model, noise, graph, dict(lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, warmup=10,
                                               grad_clip=1.0), ema_decay=0.9999, world_size=1)
    torch.manual_seed(99)
    if os.environ.get("REPRO_GRADS"):  # one backward only: which gradients differ between runs?
        x0 = torch.randint(0, cfg.tokens, (B, L), device=dev)
        tr.flat.g.zero_()
        model.train()
        loss = tr.loss_fn(model, x0)
        loss.backward()
        torch.cuda.synchronize()
        return [float(loss)], {k: v.grad.detach().clone() for k, v in model.named_parameters()}, tr.flat.g.clone()
    losses = []
    for s in range(3):
        x0 = torch.randint(0, cfg.tokens, (B, L), device=dev)
        losses.append(float(tr.train_step(x0)))
    torch.cuda.synchronize()
    global LAYOUT
    LAYOUT = [(off, p.numel(), n) for (n, p), off in zip(model.named_parameters(), tr.flat.offsets)]
    return losses, {k: v.detach().clone() for k, v in model.named_parameters()}, tr.flat.g.clone()

la, pa, ga = run()
lb, pb, gb = run()
print("losses", la, lb, "equal" if la == lb else "DIFFER")
bad = [(k, float((pa[k] - pb[k]).abs().max())) for k in pa if not torch.equal(pa[k], pb[k])]
neq = (ga != gb)
print(f"{len(bad)} of {len(pa)} parameter tensors differ bitwise; last-step gradient buffers equal: {torch.equal(ga, gb)}"
      f" ({int(neq.sum())} of {ga.numel()} elements differ, nan {int(torch.isnan(ga).sum())}; first at {int(neq.nonzero()[0]) if bool(neq.any()) else -1})")
for i in (neq.nonzero().flatten().tolist()[:8] if bool(neq.any()) else []):
    own = [(n, i - off, num) for off, num, n in LAYOUT if off <= i < off + num]
    print(f"   grad element {i}: {own or 'padding'}  {float(ga[i]):.6e} vs {float(gb[i]):.6e}")
for k, d in bad[:40]:
    print(f"   {k}: max |diff| {d:.3e}")
