"""Host-side logic that needs no GPU: LR schedule vs the reference's recorded values (golden g15), checkpoint layout
compatibility with torch.optim.AdamW / the reference's save/restore format (utils/utils.py:73-95)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR


def _g15():
    return np.load(os.path.join(GOLDEN_DIR, "g15_lr_schedule.npz"))


def test_warmup_cosine_lr_matches_reference():
    from protein_lig_sedd_amd.lr_scheduler import WarmupCosineLR, warmup_cosine_lr
    g = _g15()
    for tag in ("a", "b"):
        w, m, base, mx, mn = g[tag + "_cfg"]
        ref = g[tag + "_lrs"]
        fn = [warmup_cosine_lr(s, int(w), int(m), base, mx, mn) for s in range(len(ref))]
        np.testing.assert_allclose(fn, ref, rtol=1e-12, atol=0)
        lin = torch.nn.Linear(2, 2)
        opt = torch.optim.SGD(lin.parameters(), lr=0.1)
        sch = WarmupCosineLR(opt, int(w), int(m), base, mx, mn)
        got = [opt.param_groups[0]["lr"]]
        for _ in range(len(ref) - 1):
            opt.step()
            sch.step()
            got.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=0)
        assert sch.schedule()(3) == fn[3]


def _tiny_trainer():
    from types import SimpleNamespace as NS
    from protein_lig_sedd_amd import graph_lib, noise_lib
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.train import SEDDTrainer
    cfg = NS(model=NS(hidden_size=32, cond_dim=16, n_heads=1, n_blocks=1, dropout=0.0, scale_by_sigma=True, length=8),
             graph=NS(type="absorb"), tokens=5)
    torch.manual_seed(0)
    model = _SEDD(cfg, compute_dtype=torch.float32)
    tr = SEDDTrainer(model, noise_lib.LogLinearNoise(), graph_lib.Absorbing(5),
                     dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, warmup=10, grad_clip=1.0),
                     world_size=1)
    return model, tr


def test_checkpoint_layout_is_the_references(tmp_path):
    """optimizer entry loads into a real torch AdamW over model.parameters(); a reference-style file (AdamW state_dict
    with the noise module's extra dummy parameter, DDP `module.` prefixes) loads into the trainer."""
    from protein_lig_sedd_amd.train import restore_checkpoint, save_checkpoint
    model, tr = _tiny_trainer()
    tr.flat.m.uniform_(-1, 1)
    tr.flat.v.uniform_(0, 1)
    tr.step_count = 7
    tr.ema.num_updates = 7
    sd = tr.state_dict()
    assert set(sd) == {"optimizer", "model", "ema", "step"}
    params = list(model.parameters())
    opt = torch.optim.AdamW(params, lr=1e-3)
    opt.load_state_dict(sd["optimizer"])  # torch validates the group/param structure
    for i, p in enumerate(params):
        assert torch.equal(opt.state[p]["exp_avg"], tr.flat.view(tr.flat.m, i))
        assert torch.equal(opt.state[p]["exp_avg_sq"], tr.flat.view(tr.flat.v, i))
        assert int(opt.state[p]["step"]) == 7
    path = str(tmp_path / "ckpt" / "checkpoint.pth")
    assert restore_checkpoint(path, tr, "cpu") is tr  # missing file: unchanged, directory created
    save_checkpoint(path, tr)
    m0, v0, p0 = tr.flat.m.clone(), tr.flat.v.clone(), tr.flat.p.clone()
    model2, tr2 = _tiny_trainer()
    with torch.no_grad():
        tr2.flat.p.add_(1.0)
    restore_checkpoint(path, tr2, "cpu")
    for i in range(len(tr.flat.params)):  # (the 64-element alignment padding between tensors is not state)
        for a, b in ((tr2.flat.m, m0), (tr2.flat.v, v0), (tr2.flat.p, p0)):
            assert torch.equal(tr2.flat.view(a, i), tr.flat.view(b, i))
    assert tr2.step_count == 7 and tr2.ema.num_updates == 7
    # reference-style file: torch AdamW over model + noise parameters, DDP-prefixed model keys
    dummy = torch.nn.Parameter(torch.zeros(0))
    ref_opt = torch.optim.AdamW(params + [dummy], lr=1e-3)
    for p in params:
        p.grad = torch.randn_like(p)
    ref_opt.step()
    ref_file = {"optimizer": ref_opt.state_dict(), "model": {"module." + k: v for k, v in model.state_dict().items()},
                "ema": tr.ema.state_dict(), "step": 1}
    model3, tr3 = _tiny_trainer()
    tr3.load_state_dict(ref_file)
    for i, p in enumerate(params):
        assert torch.equal(tr3.flat.view(tr3.flat.m, i), ref_opt.state[p]["exp_avg"])
        assert torch.equal(tr3.flat.view(tr3.flat.p, i), p.detach())
    assert tr3.step_count == 1


def test_trainer_lr_rules():
    from protein_lig_sedd_amd.lr_scheduler import warmup_cosine_lr
    _, tr = _tiny_trainer()
    assert tr.current_lr(5) == pytest.approx(1e-3 * 0.5)     # losses.py:58-62 linear warm-up
    assert tr.current_lr(50) == pytest.approx(1e-3)
    tr.lr_schedule = lambda s: warmup_cosine_lr(s, 10, 50)
    assert tr.current_lr(30) == pytest.approx(warmup_cosine_lr(30, 10, 50))


def _cross_trainer(task, double_step=True):
    from types import SimpleNamespace as NS
    from protein_lig_sedd_amd import graph_lib, noise_lib
    from protein_lig_sedd_amd.model.transformers_protlig_cross import ProteinLigandDiffusionModel
    from protein_lig_sedd_amd.train import CrossAttTrainer
    cfg = NS(model=NS(hidden_size=32, cond_dim=16, n_heads=2, n_blocks=1, dropout=0.0, scale_by_sigma=True),
             graph=NS(type="absorb"))
    torch.manual_seed(0)
    model = ProteinLigandDiffusionModel(cfg)
    tr = CrossAttTrainer(model, noise_lib.LogLinearNoise(), graph_lib.Absorbing(36), graph_lib.Absorbing(2364),
                         dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, warmup=0, grad_clip=1.0),
                         task=task, double_step=double_step, world_size=1)
    return model, tr


@pytest.mark.parametrize("task", ["joint", "protein_only", "ligand_only"])
def test_cross_trainer_checkpoint_round_trip(task):
    """Adam's own step counter (2 per iteration under double_step, run_train_protlig_cross_att.py:247,325)
    survives save / load, and the optimizer state is indexed by position in model.parameters() (frozen parameters keep
    their slot, torch-style), so a task-frozen run's file maps onto the right tensors."""
    model, tr = _cross_trainer(task)
    all_params = list(model.parameters())
    trainable = [i for i, p in enumerate(all_params) if p.requires_grad]
    if task != "joint":
        assert 0 < len(trainable) < len(all_params)
    tr.flat.m.uniform_(-1, 1)
    tr.flat.v.uniform_(0, 1)
    tr.step_count, tr.opt_steps, tr.ema.num_updates = 7, 14, 7
    sd = tr.state_dict()
    assert sorted(sd["optimizer"]["state"]) == trainable
    assert all(int(e["step"]) == 14 for e in sd["optimizer"]["state"].values())
    assert sd["step"] == 7
    opt = torch.optim.AdamW(all_params, lr=1e-3)
    opt.load_state_dict(sd["optimizer"])  # torch accepts the layout: one slot per parameter, state only where trained
    for j, i in enumerate(trainable):
        assert torch.equal(opt.state[all_params[i]]["exp_avg"], tr.flat.view(tr.flat.m, j))
    _, tr2 = _cross_trainer(task)
    tr2.load_state_dict(sd)
    assert (tr2.step_count, tr2.opt_steps, tr2.ema.num_updates) == (7, 14, 7)
    for j in range(len(trainable)):
        assert torch.equal(tr2.flat.view(tr2.flat.m, j), tr.flat.view(tr.flat.m, j))
        assert torch.equal(tr2.flat.view(tr2.flat.v, j), tr.flat.view(tr.flat.v, j))
    # a file written by the reference's own optimizer: AdamW over ALL parameters + the noise module's dummy parameter,
    # stepped twice per iteration; frozen parameters never get a gradient and therefore no state
    ref_opt = torch.optim.AdamW(all_params + [torch.nn.Parameter(torch.zeros(0))], lr=1e-3)
    saved_grads = [p.grad for p in all_params]
    for i in trainable:
        all_params[i].grad = torch.randn_like(all_params[i])
    for p in all_params:
        if not p.requires_grad:
            p.grad = None
    ref_opt.step()
    ref_opt.step()
    ref_state = ref_opt.state_dict()
    for p, g in zip(all_params, saved_grads):
        p.grad = g
    ref_file = {"optimizer": ref_state, "model": model.state_dict(), "ema": tr.ema.state_dict(), "step": 1}
    _, tr3 = _cross_trainer(task)
    tr3.load_state_dict(ref_file)
    assert (tr3.step_count, tr3.opt_steps) == (1, 2)
    for j, i in enumerate(trainable):
        assert torch.equal(tr3.flat.view(tr3.flat.m, j), ref_opt.state[all_params[i]]["exp_avg"])
        assert torch.equal(tr3.flat.view(tr3.flat.v, j), ref_opt.state[all_params[i]]["exp_avg_sq"])


def test_trainer_optimizer_selection():
    """losses.py:35-46: Adam and AdamW are accepted (the kernel takes the coupled / decoupled weight-decay form), anything
    else and accum != 1 raise."""
    from types import SimpleNamespace as NS
    from protein_lig_sedd_amd import graph_lib, noise_lib
    from protein_lig_sedd_amd.model.transformer import _SEDD
    from protein_lig_sedd_amd.train import SEDDTrainer
    cfg = NS(model=NS(hidden_size=32, cond_dim=16, n_heads=1, n_blocks=1, dropout=0.0, scale_by_sigma=True, length=8),
             graph=NS(type="absorb"), tokens=5)
    mk = lambda **kw: SEDDTrainer(_SEDD(cfg, compute_dtype=torch.float32), noise_lib.LogLinearNoise(),  # noqa: E731
                                  graph_lib.Absorbing(5), dict(lr=1e-3, **kw), world_size=1)
    assert mk(optimizer="Adam", weight_decay=0.1).decoupled_wd is False
    assert mk(optimizer="AdamW").decoupled_wd is True and mk().decoupled_wd is True
    with pytest.raises(NotImplementedError):
        mk(optimizer="SGD")
    with pytest.raises(NotImplementedError):
        mk(accum=2)


def test_g1_product_noise_schedules(golden):
    """The PRODUCT noise module (protein_lig_sedd_amd/noise_lib.py — host glue on the [B] time vector, so it runs on the
    CPU as well) against the reference's recorded sigma / dsigma for all three schedules (noise_lib.py:45-94), and
    get_noise's dispatch + error (noise_lib.py:7-17)."""
    from types import SimpleNamespace as NS
    from protein_lig_sedd_amd import noise_lib
    g = golden("g1_noise")
    t = g["t"]
    cases = (("loglinear", NS(type="loglinear", eps=1e-3)), ("loglinear01", NS(type="loglinear", eps=0.1)),
             ("geometric", NS(type="geometric", sigma_min=1e-4, sigma_max=20.0)),
             ("cosine", NS(type="cosine", sigma_min=1e-3, sigma_max=1.0)))
    for name, ncfg in cases:
        noise = noise_lib.get_noise(NS(noise=ncfg))
        s, ds = noise(t)
        assert s.dtype == torch.float32 and ds.dtype == torch.float32
        np.testing.assert_allclose(s.numpy(), g[name + "_sigma"].numpy(), rtol=1e-6, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(ds.numpy(), g[name + "_dsigma"].numpy(), rtol=1e-6, atol=1e-9, err_msg=name)
        s2, _ = noise(t[:, None])  # the samplers call with t [B, 1]
        assert s2.shape == (t.numel(), 1)
    assert isinstance(noise_lib.get_noise(NS(noise=cases[2][1])).empty, torch.nn.Parameter)  # dummy parameter kept
    assert len(list(noise_lib.get_noise(NS(noise=cases[3][1])).parameters())) == 0            # Cosine has none
    with pytest.raises(ValueError):
        noise_lib.get_noise(NS(noise=NS(type="sqrt")))


def test_data_path_helpers():
    """SURVEY 8(f4) host data path: fixed-pad batches (data/tokenize.py:14,25), variable-length packing in the form the
    varlen hook takes (model/transformer.py:173-179), and DistributedSampler's shards (data/process_full_plinder.py:396-398)."""
    from torch.utils.data import DistributedSampler
    from protein_lig_sedd_amd import data
    ids, mask = data.pad_to_fixed([[5, 6, 7], list(range(10)), []], 8, pad_id=1)
    assert ids.tolist() == [[5, 6, 7, 1, 1, 1, 1, 1], list(range(8)), [1] * 8]
    assert mask.sum(1).tolist() == [3, 8, 0] and ids.dtype == torch.int64
    with pytest.raises(ValueError):
        data.pad_to_fixed([list(range(10))], 8, 1, truncation=False)
    seqs = [[1] * 5, [2] * 7, [3] * 3, [4] * 8, [5] * 2]
    tok, seqlens = data.pack_sequences(seqs, 10, pad_id=0)
    assert int(seqlens[0]) == 0 and int(seqlens.sum()) == tok.numel() and seqlens.dtype == torch.int32
    cu = seqlens.cumsum(-1).tolist()
    flat = tok.reshape(-1).tolist()
    segs = [flat[a:b] for a, b in zip(cu[:-1], cu[1:])]
    data_segs = [s for s in segs if s and s[0] != 0]
    assert sorted(map(tuple, data_segs)) == sorted(map(tuple, seqs))      # every sequence is exactly one segment
    assert all(len(set(s)) == 1 for s in segs if s)                       # padding never shares a segment with data
    assert tok.shape[0] == 3                                              # 25 tokens first-fit into rows of 10
    for n, world in ((10, 2), (11, 4), (3, 8), (64, 8)):
        for shuffle in (False, True):
            for drop_last in (False, True):
                for rank in range(world):
                    ref = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=shuffle, seed=7,
                                             drop_last=drop_last)
                    ref.set_epoch(3)
                    got = data.distributed_indices(n, rank, world, shuffle=shuffle, seed=7, epoch=3, drop_last=drop_last)
                    assert got == list(ref), (n, world, shuffle, drop_last, rank)


def test_g21_plinder_dataset_collate_split_and_shards(golden, tmp_path):
    """`ProteinLigandDataset` / `protein_ligand_collate_fn` / `split_indices` / `BatchLoader` against the
    reference's ImprovedProteinLigandDataset + improved_protein_ligand_collate_fn + random_split + DistributedSampler
    (data/process_full_plinder.py:31-243, :293-301, :396-398) on the synthetic sample list of golden g21."""
    import numpy as np
    from protein_lig_sedd_amd import data
    g = golden("g21_plinder_collate")
    for tag, n in (("a", 7), ("b", 5)):
        t = tag + "_"
        samples = []
        for i in range(n):
            samples.append({"protein_seq": str(g.z[t + "protein_seq"][i]), "ligand_smiles": str(g.z[t + "ligand_smiles"][i]),
                            "protein_coords": g.z[f"{t}pc_{i}"], "ligand_coords": g.z[f"{t}lc_{i}"],
                            "binding_value": float(g.z[t + "binding_in"][i]), "interaction_mask": g.z[f"{t}im_{i}"]})
        path = tmp_path / f"{tag}.pt"   # through a file, with the safe loader
        torch.save([{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in s.items()} for s in samples], path)
        ds = data.ProteinLigandDataset(str(path), max_protein_len=64, max_ligand_len=24)
        assert len(ds) == n
        batch = data.protein_ligand_collate_fn([ds[i] for i in range(n)])
        ref_keys = {k[len(t + "batch_"):] for k in g.keys() if k.startswith(t + "batch_")}
        assert ref_keys | {"protein_seq", "ligand_smiles"} == set(batch)       # strings: no token tensors invented
        for k in ref_keys:
            assert batch[k].dtype == g[t + "batch_" + k].dtype and torch.equal(batch[k], g[t + "batch_" + k]), k
        assert batch["protein_seq"] == [s["protein_seq"] for s in samples]
    with pytest.raises(FileNotFoundError):
        data.ProteinLigandDataset(str(tmp_path / "missing.pt"))
    # ligand-only samples, no optional fields
    ds = data.ProteinLigandDataset([{"ligand_smiles": "CCO"}, {"ligand_smiles": "c1ccccc1O"}], 16, 8)
    b = data.protein_ligand_collate_fn([ds[0], ds[1]])
    assert sorted(b) == [str(k) for k in g.z["ligonly_keys"]] and torch.equal(b["ligand_mask"], g["ligonly_mask"])
    # token-id lists: the trainer's batch keys, right-padded / truncated like the tokenizer calls
    ds = data.ProteinLigandDataset([{"protein_seq": [5, 6, 7], "ligand_smiles": list(range(3, 15))},
                                    {"protein_seq": list(range(4, 30)), "ligand_smiles": [9]}], 16, 8)
    b = data.protein_ligand_collate_fn([ds[0], ds[1]], protein_pad_id=1, ligand_pad_id=2)
    assert b["prot_tokens"].shape == (2, 16) and b["lig_tokens"].shape == (2, 8) and b["prot_tokens"].dtype == torch.int64
    assert b["prot_tokens"][0].tolist() == [5, 6, 7] + [1] * 13 and b["prot_tokens"][1].tolist() == list(range(4, 20))
    assert b["lig_tokens"][0].tolist() == list(range(3, 11)) and b["lig_tokens"][1].tolist() == [9] + [2] * 7
    assert torch.equal(b["protein_mask"], (b["prot_tokens"] != 1).float())
    cat = data.ddp_concat_batch(b["lig_tokens"], b["prot_tokens"])
    assert cat.shape == (2, 24) and int(cat[0, 8]) == 5 + 2363
    # seeded split and per-rank shards
    for n in (10, 37):
        tr, va, te = data.split_indices(n, 0.8, 0.1, 42)
        for nm, got in (("train", tr), ("val", va), ("test", te)):
            assert got == g[f"split{n}_{nm}"].tolist(), (n, nm)
        for world in (2, 3):
            for rank in range(world):
                ld = data.BatchLoader(list(range(n)), tr, 4, collate_fn=lambda x: x, rank=rank, world=world, shuffle=True)
                ld.set_epoch(2)
                assert [i for bt in ld for i in bt] == g[f"shard{n}_w{world}_r{rank}"].tolist()
                assert len(ld) == len(list(ld))
    # loaders end to end: every training sample of the split exactly once over the ranks (up to the sampler's wrap-around)
    samples = [{"protein_seq": list(range(1, 3 + i % 7)), "ligand_smiles": [3 + i]} for i in range(23)]
    seen = []
    for rank in range(2):
        trl, val, tel = data.create_data_loaders(samples, batch_size=4, max_protein_len=8, max_ligand_len=4, rank=rank, world=2)
        for bt in trl:
            assert bt["prot_tokens"].shape[1] == 8 and bt["lig_tokens"].shape[1] == 4
            seen += [int(v) - 3 for v in bt["lig_tokens"][:, 0]]
    tr, _, _ = data.split_indices(23)
    assert set(seen) == set(tr) and len(seen) == 2 * ((len(tr) + 1) // 2)


def test_param_lookup_and_mode_setting_helpers():
    """transformer._params reads the modules' own dictionaries (with the attribute route as fall-back); set_training calls
    model.train() only when some module is not in the requested mode already — same end state as the reference's
    unconditional model.train() of every step."""
    import types
    import torch.nn as nn
    from protein_lig_sedd_amd.model import layers as Lyr
    from protein_lig_sedd_amd.model import transformer as T
    ns = types.SimpleNamespace
    cfg = ns(tokens=25, graph=ns(type="absorb"),
             model=ns(hidden_size=64, cond_dim=32, n_heads=2, n_blocks=2, dropout=0.1, scale_by_sigma=True))
    m = T._SEDD(cfg)
    P = T._params(m.blocks[1], Lyr.BLOCK_PARAMS)
    assert P["mlp.0.weight"] is m.blocks[1].mlp[0].weight and P["adaLN_modulation.bias"] is m.blocks[1].adaLN_modulation.bias
    assert P["norm1.weight"] is m.blocks[1].norm1.weight and set(P) == set(Lyr.BLOCK_PARAMS)

    class Odd(nn.Module):  # a parameter reached through a plain attribute, not a registered sub-module
        def __init__(self):
            super().__init__()
            object.__setattr__(self, "inner", nn.Linear(2, 2))
    o = Odd()
    assert T._params(o, ("inner.weight",))["inner.weight"] is o.inner.weight
    calls = []
    orig = m.train
    m.train = lambda mode=True: (calls.append(mode), orig(mode))[1]
    T.set_training(m, True)
    assert calls == []                       # already in train mode everywhere: nothing to do
    m.blocks[1].eval()
    T.set_training(m, True)
    assert calls == [True] and all(mod.training for mod in m.modules())
    T.set_training(m, False)
    assert calls == [True, False] and not any(mod.training for mod in m.modules())
