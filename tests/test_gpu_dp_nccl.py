"""D1 on the one GPU a test box has: the N > 1 code path of SEDDTrainer over a 1-rank RCCL group (backend "nccl"),
bit-identical to the single-process step, for the default (one all-reduce after backward), the overlapped per-block
form and the sharded optimiser (reduce-scatter / all-gather); the bf16 gradient exchange within its rounding.  The work runs in a child process that conftest.py starts before this process initialises the GPU
(tests/dp_nccl_child.py); this test only collects its verdict."""
import json
import os
import time

import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_step_is_bit_identical(dp_nccl_child):
    proc, path = dp_nccl_child
    if proc is None:
        pytest.skip("child process was not started (no GPU visible at session start)")
    t0 = time.time()
    while proc.poll() is None and time.time() - t0 < 240:
        time.sleep(0.5)
    if proc.poll() is None:
        proc.kill()
        pytest.fail("1-rank RCCL child did not finish within 240 s")
    assert os.path.exists(path), f"child exited with {proc.returncode} and left no result"
    res = json.load(open(path))
    assert "error" not in res, res.get("error")
    for name, c in res["cases"].items():
        if "bit_identical" in c:
            assert c["bit_identical"], f"{name}: dp step differs from the single-process step: {c}"
        else:  # bf16 gradient exchange: every gradient element rounded once
            assert c["close"], f"{name}: {c}"
    # plain + overlapped (fp32, bf16 compute) | sharded optimiser | bf16 gradient exchange with and without it
    assert len(res["cases"]) == 4 + 2 + 4
    assert res["ema_survives_binding"], "a restored EMA shadow must survive the data-parallel binding"
    assert res["gloo_refused"], "init_distributed must refuse gloo for device tensors"
    assert res["ok"]
