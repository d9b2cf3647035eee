"""Does gloo + device tensors + concurrent kernel launches work on this box without any of this repo's kernels?
   (1 rank; same pattern as SEDDTrainer's overlap path: async all-reduce of 28 MB slices while matmuls keep launching.)
   Result on the MI355X box (torch 2.10 + ROCm 7.0 wheels): hangs in the first w.wait() -- plain torch matmuls, none of
   this repo's code -- whereas gloo all-reduces of device tensors with the stream idle complete (370 MB in 0.2 s).
   So gloo cannot rehearse the overlapped path on device tensors here; the rehearsals that do work are
   SEDD_DP_FORCE=1 (every collective of the N > 1 path over a 1-rank RCCL group, bench.py) and the world-size-2
   gloo tests on host tensors (tests/test_dp_cpu.py)."""
import faulthandler, os, sys, time, torch, torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
g = torch.ones(92_000_000, device="cuda")
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for it in range(5):
    pend = []
    for k in range(12):
        for _ in range(8):
            b = a @ a
        pend.append(dist.all_reduce(g[k * 7_000_000:(k + 1) * 7_000_000], async_op=True))
    for w in pend:
        w.wait()
    torch.cuda.synchronize()
    print("iter", it, "ok", flush=True)
dist.destroy_process_group()
