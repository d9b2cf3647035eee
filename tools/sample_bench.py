"""Sampling throughput at the BASELINE configs[4] per-GPU shard: cfg-2 model, Analytic predictor, batch 512 x L 1024,
bf16; times `--steps` predictor updates (model forward + fused predictor kernel) and reports sequences/s at 1024 steps.
    python tools/sample_bench.py [--batch 512] [--seq 1024] [--steps 8] [--predictor analytic]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B_
from protein_lig_sedd_amd import graph_lib, noise_lib, ops, sampling
from protein_lig_sedd_amd.model.transformer import _SEDD

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--seq", type=int, default=1024)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--predictor", default="analytic")
a = ap.parse_args()
dev = torch.device("cuda")
cfg = B_.small_config()
cfg.model.length = a.seq
model = _SEDD(cfg, compute_dtype=torch.bfloat16).to(dev).eval()
B_.randomize(model, 1234)
graph, noise = graph_lib.Absorbing(cfg.tokens), noise_lib.LogLinearNoise(eps=1e-3).to(dev)
sampler = sampling.get_pc_sampler(graph, noise, (a.batch, a.seq), a.predictor, a.steps, denoise=True, device=dev)
sampler(model)
torch.cuda.synchronize()
ops.PROF = ops.KernelProf()
t0 = time.perf_counter()
x = sampler(model)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
prof = ops.PROF.summary()
per_step = dt / (a.steps + 1)
print(json.dumps({"workload": f"{a.predictor} sampler, B={a.batch}, L={a.seq}, V={graph.dim}, bf16",
                  "ms_per_sampler_step": 1e3 * per_step, "tokens_per_s_forward": a.batch * a.seq / per_step,
                  "seqs_per_s_at_1024_steps": a.batch / (per_step * 1025),
                  "kernels_ms_per_step": {k: v[1] / (a.steps + 1) for k, v in prof.items()},
                  "masked_left": int((x == graph.dim - 1).sum())}))
