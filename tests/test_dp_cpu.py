"""world_size-2 gloo tests (CPU) of the data-parallel path: flat-buffer layout, per-block gradient buckets launched
from the backward hooks + leftover ranges cover the whole buffer exactly once, parameter broadcast, batch sharding and
the sharded-sampling all_gather.  The kernels themselves are covered by the -m gpu tests."""
import os
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    ns = types.SimpleNamespace
    return ns(tokens=25, graph=ns(type="absorb"),
              model=ns(hidden_size=64, cond_dim=32, n_heads=2, n_blocks=3, dropout=0.0, scale_by_sigma=True))


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from protein_lig_sedd_amd import graph_lib, noise_lib
        from protein_lig_sedd_amd.model.transformer import _SEDD
        from protein_lig_sedd_amd.train import SEDDTrainer, shard_batch, sharded_sampling
        torch.manual_seed(100 + rank)  # different init per rank: broadcast must make them equal
        model = _SEDD(_cfg(), compute_dtype=torch.float32)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        os.environ["SEDD_DP_OVERLAP"] = "1"  # exercise the opt-in per-block overlapped path as well as the default
        tr = SEDDTrainer(model, noise_lib.LogLinearNoise(), graph_lib.Absorbing(25), dict(lr=1e-3), world_size=world)
        assert tr.overlap
        fl = tr.flat
        # 1. broadcast: rank 0's parameters everywhere, params / grads are views of the flat buffers
        ref = [torch.empty_like(fl.p) for _ in range(world)]
        dist.all_gather(ref, fl.p)
        assert torch.equal(ref[0], ref[1])
        for i, p in enumerate(fl.params):
            assert p.data_ptr() == fl.view(fl.p, i).data_ptr() and p.grad.data_ptr() == fl.view(fl.g, i).data_ptr()
            assert fl.offsets[i] % 64 == 0
        assert torch.equal(fl.ema, fl.p)
        # 1b. resume: an adopted EMA whose shadow differs from the weights survives the binding (rank 0's shadow wins)
        from protein_lig_sedd_amd.model.ema import ExponentialMovingAverage
        m2 = _SEDD(_cfg(), compute_dtype=torch.float32)
        ema2 = ExponentialMovingAverage(m2.parameters(), 0.9999)
        with torch.no_grad():
            for sp in ema2.shadow_params:
                sp.add_(1.0 + rank)
        ema2.num_updates = 500
        os.environ["SEDD_DP_OVERLAP"] = "0"
        tr2 = SEDDTrainer(m2, noise_lib.LogLinearNoise(), graph_lib.Absorbing(25), dict(lr=1e-3), world_size=world, ema=ema2)
        both = [torch.empty_like(tr2.flat.ema) for _ in range(world)]
        dist.all_gather(both, tr2.flat.ema)
        assert torch.equal(both[0], both[1])
        i0 = next(i for i, p in enumerate(tr2.flat.params) if p.dim() == 2)
        diff = tr2.flat.view(tr2.flat.ema, i0) - tr2.flat.view(tr2.flat.p, i0)
        assert torch.allclose(diff, torch.ones_like(diff)), "restored EMA shadow was overwritten by the weights"
        assert tr2.ema.num_updates == 500 and tr2.ema.shadow_params[i0].data_ptr() == tr2.flat.view(tr2.flat.ema, i0).data_ptr()
        os.environ["SEDD_DP_OVERLAP"] = "1"
        # 2. buckets + leftovers tile [0, n) exactly once
        cover = sorted(tr._buckets + tr._leftover_ranges({(a, b) for a, b in tr._buckets}))
        pos = 0
        for lo, hi in cover:
            assert lo == pos and hi > lo
            pos = hi
        assert pos == fl.n and len(tr._buckets) == 3
        # 3. gradient all-reduce through the backward hooks (async per-block buckets) + leftovers
        fl.g.copy_(torch.arange(fl.n, dtype=torch.float32) * (rank + 1))
        for blk in reversed(model.blocks):   # the order backward finishes the blocks in
            blk._on_grads_ready()
        tr._reduce_gradients()
        expect = torch.arange(fl.n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert torch.equal(fl.g, expect)
        assert tr._pending == []
        # non-overlapped path gives the same
        tr.overlap = False
        fl.g.copy_(torch.arange(fl.n, dtype=torch.float32) * (rank + 1))
        tr._reduce_gradients()
        assert torch.equal(fl.g, expect)
        # 3b. the cheaper exchanges: bf16 gradients and / or reduce-scatter onto this rank's shard (zero1), then the
        #     all-gather that follows the sharded optimiser — every rank must end with the same whole buffers
        os.environ["SEDD_DP_OVERLAP"] = "0"
        for comm_dt in (torch.float32, torch.bfloat16):
            for z1 in (False, True):
                m3 = _SEDD(_cfg(), compute_dtype=torch.float32)
                tr3 = SEDDTrainer(m3, noise_lib.LogLinearNoise(), graph_lib.Absorbing(25), dict(lr=1e-3), world_size=world,
                                  grad_comm_dtype=comm_dt, zero1=z1)
                f3 = tr3.flat
                assert tr3.zero1 == z1 and f3.n % (64 * (world if z1 else 1)) == 0
                lo, hi = tr3.shard
                assert (hi - lo) * (world if z1 else 1) == f3.n and lo == (rank * (hi - lo) if z1 else 0)
                base = ((torch.arange(f3.n, dtype=torch.float32) % 97) - 48.0) / 64.0   # bf16-exact values
                f3.g.copy_(base * (rank + 1))
                tr3._exchange_gradients()
                want = base * sum(r + 1 for r in range(world))
                assert torch.equal(f3.g[lo:hi], want[lo:hi]), (comm_dt, z1)
                if not z1:
                    assert torch.equal(f3.g, want)
                else:
                    # what the sharded optimiser leaves: each rank's own slice of p / m / v / ema is new, the rest stale
                    for flat in (f3.p, f3.m, f3.v, f3.ema):
                        flat.zero_()
                        flat[lo:hi] = torch.arange(lo, hi, dtype=torch.float32) + 1
                    tr3._gather_flat(f3.p)
                    assert torch.equal(f3.p, torch.arange(f3.n, dtype=torch.float32) + 1)
                    tr3._shards_dirty = True
                    sd3 = tr3.ema.state_dict()          # reading the EMA as a whole gathers the shards first
                    assert torch.equal(f3.ema, torch.arange(f3.n, dtype=torch.float32) + 1) and not tr3._shards_dirty
                    assert torch.equal(f3.m, f3.ema) and torch.equal(f3.v, f3.ema) and len(sd3["shadow_params"]) == len(f3.params)
        os.environ["SEDD_DP_OVERLAP"] = "1"
        # 4. scalar loss all-reduce(AVG)
        assert abs(float(tr.reduce_loss(torch.tensor(float(rank)))) - 0.5) < 1e-6
        # 5. batch sharding + sharded sampling gather
        batch = torch.arange(8 * 5).view(8, 5)
        mine = shard_batch(batch, rank, world)
        assert torch.equal(mine, batch[rank * 4:(rank + 1) * 4])
        out = sharded_sampling(lambda m, **kw: mine + 0, None, rank, world)
        assert torch.equal(out, batch)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_dp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_lr_warmup_and_ema_decay_host_math():
    sys.path.insert(0, ROOT)
    from protein_lig_sedd_amd.model.ema import ExponentialMovingAverage
    e = ExponentialMovingAverage([], 0.9999)
    assert abs(e.current_decay() - 2 / 11) < 1e-12
    from protein_lig_sedd_amd.train import SEDDTrainer
    t = SEDDTrainer.__new__(SEDDTrainer)
    t.lr, t.warmup = 3e-4, 100
    assert t.current_lr(1) == 3e-4 / 100 and t.current_lr(100) == 3e-4 and t.current_lr(1000) == 3e-4
