"""Headline benchmark: train tokens/s of the SEDD hot path on N MI355X GPUs (BASELINE.json `metric`), one process per
GPU (torch.distributed.run, RCCL).  Workload = BASELINE.json configs[1]: SEDD small (D=768, 12 blocks, 12 heads,
cond 128, ~92 M params), absorbing graph, 25-token AA vocab (V=26), L=512, bf16 compute, per-GPU batch 64, AdamW +
warm-up + clip 1.0 + EMA, dropout 0.1 — synthetic token ids, random-init weights (N(0, 0.02) also for the reference's
zero-initialised adaLN / output weights so every path is numerically live).

A step = one full train step (corrupt -> forward -> score-entropy loss -> backward -> gradient all-reduce ->
clip/AdamW/EMA).  Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md chip table)
PEAK_HBM_GBS = 8000.0


def ns(**kw):
    return types.SimpleNamespace(**kw)


def small_config(dropout=0.1):
    return ns(tokens=25, graph=ns(type="absorb"),
              model=ns(hidden_size=768, cond_dim=128, n_heads=12, n_blocks=12, dropout=dropout, scale_by_sigma=True))


def randomize(model, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2 or name.endswith("bias"):
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))


def cpu_baseline(seconds_budget=12.0):
    """The CPU oracle (oracle/sedd_cpu.py, kind "port": a plain-PyTorch fp32 restatement of the reference's train step)
    timed on this host's cores on a bounded sample of the same workload: same model, L=512, batch 2."""
    from oracle import sedd_cpu as O
    from oracle.weights import seeded_state_dict
    torch.manual_seed(0)
    D, H, C, nb, tokens, B, L = 768, 12, 128, 12, 25, 2, 512
    shapes = {"vocab_embed.embedding.weight": (tokens + 1, D), "sigma_map.mlp.0.weight": (C, 256),
              "sigma_map.mlp.0.bias": (C,), "sigma_map.mlp.2.weight": (C, C), "sigma_map.mlp.2.bias": (C,),
              "output_layer.norm_final.weight": (D,), "output_layer.linear.weight": (tokens + 1, D),
              "output_layer.linear.bias": (tokens + 1,), "output_layer.adaLN_modulation.weight": (2 * D, C),
              "output_layer.adaLN_modulation.bias": (2 * D,)}
    for i in range(nb):
        for k, v in {"norm1.weight": (D,), "norm2.weight": (D,), "attn_qkv.weight": (3 * D, D), "attn_out.weight": (D, D),
                     "mlp.0.weight": (4 * D, D), "mlp.0.bias": (4 * D,), "mlp.2.weight": (D, 4 * D), "mlp.2.bias": (D,),
                     "adaLN_modulation.weight": (6 * D, C), "adaLN_modulation.bias": (6 * D,)}.items():
            shapes[f"blocks.{i}.{k}"] = v
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(shapes, seed=0).items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    ema = {k: v.detach().clone() for k, v in sd.items()}
    graph, noise = O.AbsorbingCPU(tokens), O.make_noise("loglinear", eps=1e-3)

    def step(s):
        x0 = torch.randint(0, tokens, (B, L))
        t = torch.rand(B) * 0.999 + 1e-3
        xt = graph.sample_transition(x0, noise(t)[0][:, None], torch.rand(B, L))
        for p in sd.values():
            p.grad = None
        loss, _ = O.sedd_loss(sd, graph, noise, x0, t, xt, H, nb)
        loss.backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in sd.values()))
        coef = O.clip_coef(gn, 1.0)
        with torch.no_grad():
            for k, p in sd.items():
                O.adamw_step(p, p.grad, m[k], v2[k], s, 3e-4, clip_coef=coef)
                O.ema_update(ema[k], p, O.ema_decay(s, 0.9999))

    step(1)  # warm-up
    t0, times = time.perf_counter(), []
    while len(times) < 5 or (time.perf_counter() - t0 < seconds_budget and len(times) < 32):  # median of >= 5
        t1 = time.perf_counter()
        step(len(times) + 2)
        times.append(time.perf_counter() - t1)
    med = sorted(times)[len(times) // 2]
    return {"value": B * L / med, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {len(times)} train steps (after 1 warm-up) of the same model at batch {B} x L {L}, fp32, "
                      "oracle/sedd_cpu.py + AdamW/EMA",
            "step_s_min_median_max": [min(times), med, max(times)],
            "cfg1": cpu_baseline_cfg1()}


def cpu_baseline_cfg1(euler_steps_timed=12):
    """BASELINE configs[0] — the reference's own CPU-runnable case (tiny SEDD: D768 x 8 blocks, 8 heads of 96, L=128,
    batch 8, V=26, fp32): one oracle train step (fwd + bwd + clip + AdamW + EMA) and the 128-step Euler sampler + denoise
    (SURVEY 8d), on all host threads torch uses here and on 8 threads (the survey container measured 1.19 s / 34.9 s on
    8 cores with the reference itself).  The sampler is timed on a bounded sample of its 129 model evaluations."""
    from oracle import sedd_cpu as O
    from oracle.weights import seeded_state_dict
    D, H, C, nb, tokens, B, L = 768, 8, 128, 8, 25, 8, 128
    shapes = {"vocab_embed.embedding.weight": (tokens + 1, D), "sigma_map.mlp.0.weight": (C, 256),
              "sigma_map.mlp.0.bias": (C,), "sigma_map.mlp.2.weight": (C, C), "sigma_map.mlp.2.bias": (C,),
              "output_layer.norm_final.weight": (D,), "output_layer.linear.weight": (tokens + 1, D),
              "output_layer.linear.bias": (tokens + 1,), "output_layer.adaLN_modulation.weight": (2 * D, C),
              "output_layer.adaLN_modulation.bias": (2 * D,)}
    for i in range(nb):
        for k, v in {"norm1.weight": (D,), "norm2.weight": (D,), "attn_qkv.weight": (3 * D, D), "attn_out.weight": (D, D),
                     "mlp.0.weight": (4 * D, D), "mlp.0.bias": (4 * D,), "mlp.2.weight": (D, 4 * D), "mlp.2.bias": (D,),
                     "adaLN_modulation.weight": (6 * D, C), "adaLN_modulation.bias": (6 * D,)}.items():
            shapes[f"blocks.{i}.{k}"] = v
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(shapes, seed=0).items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    ema = {k: v.detach().clone() for k, v in sd.items()}
    graph, noise = O.AbsorbingCPU(tokens), O.make_noise("loglinear", eps=1e-3)

    def train_step(s):
        x0 = torch.randint(0, tokens, (B, L))
        t = torch.rand(B) * 0.999 + 1e-3
        xt = graph.sample_transition(x0, noise(t)[0][:, None], torch.rand(B, L))
        for p in sd.values():
            p.grad = None
        loss, _ = O.sedd_loss(sd, graph, noise, x0, t, xt, H, nb)
        loss.backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in sd.values()))
        coef = O.clip_coef(gn, 1.0)
        with torch.no_grad():
            for k, p in sd.items():
                O.adamw_step(p, p.grad, m[k], v2[k], s, 3e-4, clip_coef=coef)
                O.ema_update(ema[k], p, O.ema_decay(s, 0.9999))

    def euler(nsteps_timed, total=128):
        x = graph.sample_limit(B, L)
        ts = torch.linspace(1, 1e-5, total + 1)
        dt = (1 - 1e-5) / total
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(nsteps_timed):
                sg = noise(ts[i] * torch.ones(B, 1))
                x = O.euler_step(graph, O.sedd_forward(sd, x, sg[0].reshape(-1), H, nb), x, sg, dt, torch.rand(B, L, tokens + 1))
        return (time.perf_counter() - t0) / nsteps_timed

    out = {"workload": "BASELINE configs[0]: SEDD tiny D768 x 8 blocks, 8 heads, L=128, batch 8, V=26, fp32"}
    default_threads = torch.get_num_threads()
    for nthr in sorted({default_threads, min(8, default_threads)}, reverse=True):
        torch.set_num_threads(nthr)
        train_step(1)
        tt = []
        for s in range(5):
            t0 = time.perf_counter()
            train_step(s + 2)
            tt.append(time.perf_counter() - t0)
        ts_ = sorted(tt)[2]
        per_eval = euler(euler_steps_timed if nthr == default_threads else 4)
        out[f"threads_{nthr}"] = {"train_step_s": ts_, "train_tokens_per_s": B * L / ts_,
                                  "euler_128_steps_plus_denoise_s": per_eval * 129, "sampled_seqs_per_s": B / (per_eval * 129),
                                  "sample": "median of 5 train steps after 1 warm-up; sampler = per-evaluation time of the first Euler steps x 129"}
    torch.set_num_threads(default_threads)
    return out


# average HBM bytes per bf16 GEMM launch of this workload (B=64, L=512), measured with the PMC counters as the guide
# prescribes (separate --pmc passes; FETCH_SIZE doubled on gfx950): profiles/r3f_pmc_hbm_traffic.csv, last row (160 launches
# per step; 279.4e6 in r2_pmc_hbm_traffic.csv, when the family still held the 26 small adaLN-backward launches of a step)
GEMM_HBM_BYTES_PER_LAUNCH = 322.7e6


def gpu_cfg1(dev):
    """BASELINE configs[0] on the GPU, beside the CPU figures above: the same tiny model (fp32, as on the CPU), one train
    step and the whole 128-step Euler sampler + denoise — eager, and with the predictor update replayed as a HIP graph."""
    from protein_lig_sedd_amd import graph_lib, noise_lib, sampling
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.train import SEDDTrainer
    ns = types.SimpleNamespace
    B, L, steps = 8, 128, 128
    cfg = ns(tokens=25, graph=ns(type="absorb"),
             model=ns(hidden_size=768, cond_dim=128, n_heads=8, n_blocks=8, dropout=0.1, scale_by_sigma=True))
    model = _SEDD(cfg, compute_dtype=torch.float32).to(dev)
    randomize(model, 1)
    graph, noise = graph_lib.Absorbing(25), noise_lib.LogLinearNoise(eps=1e-3).to(dev)
    out = {}
    for hg in (False, True):
        fn = sampling.get_pc_sampler(graph, noise, (B, L), "euler", steps, denoise=True, eps=1e-5, device=dev, hip_graph=hg)
        fn(model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(model)
        torch.cuda.synchronize()
        out["euler_128_steps_plus_denoise_s" + ("_hip_graph" if hg else "")] = time.perf_counter() - t0
    tr = SEDDTrainer(model, noise, graph, dict(lr=3e-4, warmup=100, grad_clip=1.0), ema_decay=0.9999, world_size=1)
    x = torch.randint(0, 25, (B, L), device=dev)
    for _ in range(3):
        tr.train_step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(x)
    torch.cuda.synchronize()
    out["train_step_s"] = (time.perf_counter() - t0) / 10
    out["train_tokens_per_s"] = B * L / out["train_step_s"]
    # the same step replayed as ONE HIP graph (SEDDTrainer(hip_graph=True): bit-identical results, tested): at 1024 tokens
    # per step the eager step is bound by the host's launch rate
    for cdt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        m2 = _SEDD(cfg, compute_dtype=cdt).to(dev)
        randomize(m2, 1)
        for hg in (False, True):
            tr2 = SEDDTrainer(m2, noise, graph, dict(lr=3e-4, warmup=100, grad_clip=1.0), ema_decay=0.9999, world_size=1,
                              hip_graph=hg)
            for _ in range(5):
                tr2.train_step(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                tr2.train_step(x)
            torch.cuda.synchronize()
            out[f"train_step_s_{tag}" + ("_hip_graph" if hg else "")] = (time.perf_counter() - t0) / 20
    out["sampled_seqs_per_s"] = B / out["euler_128_steps_plus_denoise_s"]
    return out


def hbm_kernels_cfg5(graph, dev, B=512, L=1024, iters=20):
    """the HBM-bound loss / sampling kernels at the BASELINE configs[4] per-GPU shard size (512 x 1024 rows, V = 26,
    bf16 log-scores): algorithmic bytes / HIP-event time on the launch stream."""
    from protein_lig_sedd_amd import ops, _lib as L_
    V = graph.dim
    M = B * L
    x = torch.randint(0, V, (B, L), device=dev)
    x[torch.rand(B, L, device=dev) < 0.5] = V - 1
    x0 = torch.randint(0, V - 1, (B, L), device=dev)
    logits = torch.randn(B, L, V, device=dev).to(torch.bfloat16)
    u = torch.rand(B, L, V, device=dev)
    sig = torch.rand(B, device=dev) + 0.1
    out = {}
    for name, nbytes, fn in (
            ("predictor_step_analytic", M * (V * 2 + V * 4 + 16), lambda: ops.predictor_step(graph, L_.STEP_ANALYTIC, logits, x, sig, u)),
            ("predictor_step_euler", M * (V * 2 + V * 4 + 16), lambda: ops.predictor_step(graph, L_.STEP_EULER, logits, x, sig, u)),
            # the two ways a sampler step gets its noise (SURVEY 8d "report both"): torch.rand + the kernel that reads it
            # (bytes: the rand kernel's write included), and the kernel generating Philox4x32-10 noise itself
            ("rand_plus_predictor_step_analytic", M * (V * 2 + 2 * V * 4 + 16),
             lambda: ops.predictor_step(graph, L_.STEP_ANALYTIC, logits, x, sig, torch.rand(B, L, V, device=dev))),
            ("predictor_step_analytic_philox", M * (V * 2 + 16),
             lambda: ops.predictor_step_philox(graph, L_.STEP_ANALYTIC, logits, x, sig, ops.PhiloxNoise(1234, 7))),
            ("score_entropy_fwd_bwd", M * (2 * V * 2 + 20), lambda: ops.score_entropy_fwd(graph, logits, sig, x, x0, gw=sig))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        out[name] = {"avg_us": us, "achieved_GBs": nbytes / us / 1e3, "frac_hbm_peak": nbytes / us / 1e3 / PEAK_HBM_GBS}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sample-steps", type=int, default=4, help="Analytic sampler steps timed for the seqs/s side metric")
    ap.add_argument("--sample-batch", type=int, default=512, help="per-GPU sampling shard (BASELINE configs[4]: 4096 / 8)")
    ap.add_argument("--sample-seq", type=int, default=1024, help="sampled sequence length (BASELINE configs[4])")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON: native libraries (RCCL prints a version banner at communicator
    # creation) write to file descriptor 1 directly, so fd 1 is pointed at stderr for the run and the JSON goes to
    # the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("SEDD_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SEDD_BENCH_WATCHDOG"]), exit=True)

    from protein_lig_sedd_amd import graph_lib, noise_lib, ops, sampling
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.train import SEDDTrainer, init_distributed

    rank, local, world = init_distributed()
    if world != args.gpus and rank == 0:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.manual_seed(1234)  # identical weights on every rank
    cfg = small_config()
    model = _SEDD(cfg, compute_dtype=torch.bfloat16).to(dev)
    randomize(model, 1234)
    nparams = sum(p.numel() for p in model.parameters())
    graph, noise = graph_lib.Absorbing(cfg.tokens), noise_lib.LogLinearNoise(eps=1e-3).to(dev)
    trainer = SEDDTrainer(model, noise, graph, dict(lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                                                    warmup=2500, grad_clip=1.0), ema_decay=0.9999, world_size=world)
    B, L = args.batch, args.seq
    torch.manual_seed(1234 + rank)  # per-rank data / noise
    batches = [torch.randint(0, cfg.tokens, (B, L), device=dev) for _ in range(4)]

    multi = dist.is_initialized()  # world > 1, or the 1-rank rehearsal of the collective path (SEDD_DP_FORCE=1)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.train_step(batches[i % 4])
    sync()
    trainer.reset_allreduce_timing()
    # HIP events on the launch stream around every 5th launch of each kernel family (the sampling phase moves on by one
    # every 97 launches, so over the timed steps every launch position of a step is sampled about equally often whatever
    # the number of launches per step is); bracketing every launch costs 1.2 ms per step (tools/event_overhead.py)
    ops.PROF = ops.KernelProf(stride=5)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.train_step(batches[i % 4])
    sync()
    dt = time.perf_counter() - t0
    prof = ops.PROF.summary()
    ops.PROF = None
    allreduce_ms = trainer.allreduce_ms_per_step()  # None unless the data-parallel path ran
    if multi:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    tokens_per_s = world * B * L * args.steps / dt

    # beside the headline (not part of it): the same step without the roofline's timing events, and with the weight
    # gradients on the side stream (SEDDTrainer(wgrad_stream=True): same bits; off in the timed region because kernels of
    # two streams running side by side would make the per-launch event times meaningless)
    variants = None
    if not multi:
        def steps_ms(n):
            sync()
            ta = time.perf_counter()
            for i in range(n):
                trainer.train_step(batches[i % 4])
            sync()
            return 1e3 * (time.perf_counter() - ta) / n
        variants = {"ms_per_step_no_events": steps_ms(args.steps)}
        trainer._side, trainer._side_max = ops.SideStream(dev), None
        steps_ms(2)
        variants["ms_per_step_wgrad_stream"] = steps_ms(args.steps)
        trainer._side = None

    # second half of the metric: sampled sequences/s at BASELINE configs[4] — the cfg-2 model, Analytic predictor, 1024
    # steps, shape (4096, 1024) sharded 512 per GPU (sampling/run_sample_cond.py:40-42) — measured on `--sample-steps`
    # timed sampler steps (+ the denoiser step) of exactly that per-GPU shard and scaled to the 1025 model evaluations of
    # a full run; every step does the same work (bf16 model forward on 512 x 1024 tokens + one fused predictor kernel)
    samp = None
    if args.sample_steps > 0:
        model.eval()
        SB, SL = args.sample_batch, args.sample_seq
        sampler = sampling.get_pc_sampler(graph, noise, (SB, SL), "analytic", args.sample_steps, denoise=True, device=dev)
        warm = sampling.get_pc_sampler(graph, noise, (SB, SL), "analytic", 1, denoise=True, device=dev)
        warm(model)
        sync()
        ops.PROF = ops.KernelProf(stride=1 if args.sample_steps <= 16 else 53)
        t1 = time.perf_counter()
        sampler(model)
        sync()
        ds = time.perf_counter() - t1
        sprof = ops.PROF.summary()
        ops.PROF = None
        if multi:
            tt = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ds = float(tt)
        evals = args.sample_steps + 1
        full_run = args.sample_steps >= 1024  # a whole run timed end to end: nothing is scaled
        n_l, ms_l, by_l = sprof.get("predictor_step", (0, 0.0, 0.0))
        n_sg, ms_sg, fl_sg = sprof.get("gemm_bf16_nt", (0, 0.0, 0.0))
        n_sa, ms_sa, fl_sa = sprof.get("attn_fwd", (0, 0.0, 0.0))
        hbm = hbm_kernels_cfg5(graph, dev)
        samp = {"workload": "BASELINE configs[4]: cfg-2 model, Analytic predictor, 1024 steps, per-GPU shard %d x %d, "
                            "V=26, bf16 forward; " % (SB, SL) + (
                                "the whole %d-step run + denoise timed end to end (measured, not scaled)" % args.sample_steps
                                if full_run else
                                "timed on %d sampler steps + denoise, scaled to 1025 evaluations" % args.sample_steps),
                "measured_full_run": full_run, "sampler_wall_s": ds,
                "ms_per_sampler_step": 1e3 * ds / evals,
                "seqs_per_s_at_1024_steps": world * SB / ds if full_run else world * SB / (ds / evals * 1025),
                "forward_tokens_per_s": world * SB * SL * evals / ds,
                "gemm_ms_per_step": ms_sg / evals, "gemm_TFLOPs": fl_sg / max(ms_sg, 1e-9) / 1e9,
                "attn_ms_per_step": ms_sa / evals, "attn_TFLOPs": fl_sa / max(ms_sa, 1e-9) / 1e9,
                "hbm_kernels_at_cfg5_shard": hbm,
                "predictor_step": {"launches": n_l, "avg_us": 1e3 * ms_l / max(n_l, 1),
                                   "achieved_GBs": by_l / max(ms_l, 1e-9) / 1e6,
                                   "frac_hbm_peak": by_l / max(ms_l, 1e-9) / 1e6 / PEAK_HBM_GBS}}

    if rank == 0:
        n_g, ms_g, fl_g = prof.get("gemm_bf16_nt", (0, 0.0, 0.0))
        ach = fl_g / max(ms_g, 1e-9) / 1e9  # TFLOP/s
        others = {k: {"launches": v[0], "ms_per_step": v[1] / args.steps,
                      "rate": v[2] / max(v[1], 1e-9) / 1e9} for k, v in prof.items()}
        out = {
            "metric": "train tokens/sec", "value": tokens_per_s, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SEDD small (D768 x 12 blocks, 12 heads, V=26), L=512, "
                                   "bf16, per-GPU batch %d, AdamW+EMA, dropout 0.1, dp%d" % (B, world),
                       "global_batch": B * world, "seq_len": L, "params": nparams, "parallelism": f"dp{world}"},
            "loss": float(loss),
            "dp": {"rccl_world_size": dist.get_world_size() if multi else 1, "backend": dist.get_backend() if multi else None,
                   "grad_bytes_per_step": 4 * trainer.flat.n, "allreduce_ms_per_step": allreduce_ms,
                   "form": "per-block overlapped buckets" if trainer.overlap else "one all-reduce of the flat gradient buffer after backward"},
            "roofline": {"bound": "mfma",
                         "kernel": "gemm_bf16_big_kernel NT/TN (+ gemm_bf16_nt_glds_kernel for small shapes): every "
                                   "Linear fwd / dgrad / wgrad launch of the step",
                         "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                         "traffic": GEMM_HBM_BYTES_PER_LAUNCH if (B, L) == (64, 512) else None,
                         "traffic_source": "profiles/r3f_pmc_hbm_traffic.csv (2 x FETCH_SIZE + WRITE_SIZE, separate "
                                           "rocprofv3 --pmc passes of this command, average over the bf16 GEMM launches)",
                         "launches_per_step": n_g / max(args.steps, 1),
                         "timed_sample": "HIP events around every 5th launch of the family, scaled to all launches",
                         "avg_launch_us": 1e3 * ms_g / max(n_g, 1),
                         "gemm_ms_per_step": ms_g / max(args.steps, 1)},
            "kernels": others,
        }
        if variants:
            out["variants"] = variants
        if samp:
            out["sampling"] = samp
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["cfg1"]["this_gpu"] = gpu_cfg1(dev)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
