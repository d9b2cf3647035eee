"""N-rank rehearsal of SEDDTrainer on one GPU (gloo moves the bytes): stage-by-stage progress per rank.
   SEDD_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_rehearse.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from protein_lig_sedd_amd import graph_lib, noise_lib
from protein_lig_sedd_amd.model.transformer import _SEDD
from protein_lig_sedd_amd.train import SEDDTrainer, init_distributed

rank, local, world = init_distributed()
os.makedirs("gpurun_out", exist_ok=True)
log = open(f"gpurun_out/dist_rank{rank}.log", "w")
def P(*a):
    print(f"[{time.time():.2f}] rank {rank}:", *a, file=log, flush=True)
P("init done, world", world)
import faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("REHEARSE_WATCHDOG", "60")), exit=True, file=log)
ns = types.SimpleNamespace
BIG = os.environ.get("REHEARSE_BIG", "0") == "1"   # BASELINE configs[1] sized model (D768, 12 blocks, L512)
D, H, NB, L, C = (768, 12, 12, 512, 128) if BIG else (256, 4, 3, 128, 128)
cfg = ns(tokens=25, graph=ns(type="absorb"), model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=NB, dropout=0.1, scale_by_sigma=True, length=L))
dev = torch.device("cuda", local)
torch.manual_seed(1)
model = _SEDD(cfg, compute_dtype=torch.bfloat16).to(dev)
P("model built")
tr = SEDDTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(dev), graph_lib.Absorbing(25),
                 dict(lr=3e-4, warmup=10, grad_clip=1.0), world_size=world)
P("trainer built, buckets", len(tr._buckets), "overlap", tr.overlap)
torch.manual_seed(100 + rank)
_reduce = tr._reduce_gradients
def traced_reduce():
    if os.environ.get("REHEARSE_SYNC", "0") == "1":
        torch.cuda.synchronize()
        P("  backward finished on device; reducing")
    _reduce()
    P("  reduce returned")
tr._reduce_gradients = traced_reduce
_loss_fn = tr.loss_fn
def traced_loss(*a, **k):
    P("  forward start")
    out = _loss_fn(*a, **k)
    P("  forward launched")
    return out
tr.loss_fn = traced_loss
for step in range(3):
    x = torch.randint(0, 25, (8, L), device=dev)
    loss = tr.train_step(x)
    torch.cuda.synchronize()
    P("step", step, "loss", float(loss), "gnorm", tr.grad_norm())
# replicas must stay bit-identical
chk = tr.flat.p.double().sum().reshape(1)
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
P("param checksums", [float(b) for b in both])
assert all(float(b) == float(both[0]) for b in both)
P("avg loss", float(tr.reduce_loss(loss)))
dist.barrier()
P("OK")
dist.destroy_process_group()
