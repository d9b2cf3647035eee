"""BASELINE configs[2] (cfg 3: dual-track cross-attention model of configs/config.yaml — D 1024, 32 heads, 12 blocks per
track, cond 128, Lp 512 + Ll 128, V 37 / 2365, dropout 0.1, joint task, batch 4) train step time on one GPU, in the
reference's fp32 (its autocast dtype is float32) and in bf16.  CFG3_BATCH overrides the batch, CFG3_DTYPES the list."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_lig_sedd_amd import graph_lib, noise_lib
from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
from protein_lig_sedd_amd.train import CrossAttTrainer
ns = types.SimpleNamespace
dev = torch.device("cuda", 0)
D, H, C, nb = 1024, 32, 128, 12
Lp, Ll = int(os.environ.get("CFG3_LP", "512")), int(os.environ.get("CFG3_LL", "128"))
B = int(os.environ.get("CFG3_BATCH", "4"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name in os.environ.get("CFG3_DTYPES", "fp32,bf16").split(","):
    dt = dict(fp32=torch.float32, bf16=torch.bfloat16)[name]
    cfg = ns(graph=ns(type="absorb"), model=ns(hidden_size=D, cond_dim=C, n_heads=H, n_blocks=nb, dropout=0.1, scale_by_sigma=True))
    torch.manual_seed(0)
    model = ProteinLigandDiffusionModel(cfg, compute_dtype=dt).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():  # adaLN / final layers are zero-initialised: give every weight a value so no kernel sees zeros
        for p in model.parameters():
            if p.dim() >= 2: p.copy_(torch.randn(p.shape, device=dev, generator=g) * 0.02)
    tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(eps=1e-3).to(dev), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                         dict(lr=1e-4, warmup=1000, grad_clip=1.0), task="joint", ema_decay=0.9999, world_size=1,
                         hip_graph=bool(os.environ.get("CFG3_GRAPH")), wgrad_stream=os.environ.get("CFG3_SIDE", "1") == "1")
    bs = [dict(prot_tokens=torch.randint(0, 36, (B, Lp), device=dev), lig_tokens=torch.randint(0, 2364, (B, Ll), device=dev)) for _ in range(4)]
    for i in range(4): tr.train_step(bs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get("CFG3_CPROFILE"):  # where the host's enqueue time goes (the step is host-bound at batch 4)
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for i in range(n): out = tr.train_step(bs[i % 4])
        pr.disable()
        st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45); st.sort_stats("cumulative").print_stats(60)
    elif os.environ.get("CFG3_SYNC_EACH"):  # every step starts on an idle GPU: host time of one step without back-pressure
        hs, ts = [], []
        for i in range(n):
            torch.cuda.synchronize(); a0 = time.perf_counter()
            out = tr.train_step(bs[i % 4]); a1 = time.perf_counter()
            torch.cuda.synchronize(); a2 = time.perf_counter()
            hs.append(a1 - a0); ts.append(a2 - a0)
        print(f"  per-step with a sync between steps: host enqueue {1e3*sum(hs)/n:.2f} ms, step {1e3*sum(ts)/n:.2f} ms", flush=True)
    else:
        for i in range(n): out = tr.train_step(bs[i % 4])
    host = (time.perf_counter() - t0) / n  # time to ENQUEUE a step: ~ the step time when the host is the limiter
    torch.cuda.synchronize()
    s = (time.perf_counter() - t0) / n
    P = sum(p.numel() for p in model.parameters())
    print(f"cfg3 {name} batch {B}{' hip_graph' if os.environ.get('CFG3_GRAPH') else ''}{' single-stream' if os.environ.get('CFG3_SIDE', '1') != '1' else ''}: {s*1e3:.2f} ms/step, {B*(Lp+Ll)/s:.0f} tokens/s ({B/s:.1f} pairs/s), params {P/1e6:.1f} M, loss {float(out[0]):.4f}, host enqueue {host*1e3:.2f} ms/step", flush=True)
    del tr, model
    torch.cuda.empty_cache()
