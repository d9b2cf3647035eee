"""BASELINE configs[0] (tiny SEDD, 25 tokens, L 128, batch 8) on the GPU: one train step and the 128-step Euler sampler,
eager and with the predictor update replayed as a HIP graph."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from protein_lig_sedd_amd import graph_lib, noise_lib, sampling
from protein_lig_sedd_amd.model.transformer import _SEDD
from protein_lig_sedd_amd.train import SEDDTrainer
ns = types.SimpleNamespace
dev = torch.device("cuda", 0)
for name, (D, H, C, nb) in dict(tiny=(768, 8, 128, 8)).items():
    for cdt in (torch.float32, torch.bfloat16):
        cfg = ns(tokens=25, graph=ns(type="absorb"), model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=nb, dropout=0.1, scale_by_sigma=True))
        model = _SEDD(cfg, compute_dtype=cdt).to(dev)
        bench.randomize(model, 1)
        graph, noise = graph_lib.Absorbing(25), noise_lib.LogLinearNoise(eps=1e-3).to(dev)
        B, L, steps = 8, 128, 128
        res = []
        for hg in (False, True):
            fn = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, denoise=True, eps=1e-5, device=dev, hip_graph=hg)
            fn(model); torch.cuda.synchronize()
            t0 = time.perf_counter(); out = fn(model); torch.cuda.synchronize()
            res.append(time.perf_counter() - t0)
        tr = SEDDTrainer(model, noise, graph, dict(lr=3e-4, warmup=100, grad_clip=1.0), ema_decay=0.9999, world_size=1)
        x = torch.randint(0, 25, (B, L), device=dev)
        for _ in range(3): tr.train_step(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tr.train_step(x)
        torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 10
        print(f"{name} (D{D}x{nb}) {str(cdt)[6:]}: 128-step Euler + denoise, batch 8 x L 128: eager {res[0]*1e3:.1f} ms ({res[0]/129*1e3:.2f} ms/step), "
              f"HIP graph {res[1]*1e3:.1f} ms ({res[1]/129*1e3:.2f} ms/step); train step {ts*1e3:.2f} ms", flush=True)
