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
