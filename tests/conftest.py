import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_DP_CHILD = {"proc": None, "path": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tests/test_gpu_dp_nccl.py needs a FRESH process for its 1-rank RCCL group.  A process that has initialised the
    # GPU must not start programs on this pool, so the child is started here, before any test touches the device
    # (torch.cuda.device_count() does not initialise it), and the test collects the result later.
    mark = config.getoption("-m", default="") or ""
    if "gpu" in mark and "not gpu" not in mark and torch.cuda.device_count() > 0 and not os.environ.get("SEDD_NO_DP_CHILD"):
        import subprocess
        import tempfile
        path = os.path.join(tempfile.mkdtemp(prefix="sedd_dp_"), "result.json")
        log = open(path + ".log", "w")
        _DP_CHILD["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_nccl_child.py"), path],
                                             stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
        _DP_CHILD["path"] = path


@pytest.fixture(scope="session")
def dp_nccl_child():
    return _DP_CHILD["proc"], _DP_CHILD["path"]


class Golden:
    """npz fixture produced by oracle/gen_golden.py from the imported reference."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, k):
        a = self.z[k]
        return torch.from_numpy(a) if a.ndim > 0 else torch.tensor(a.item())

    def keys(self):
        return list(self.z.keys())

    def sub(self, prefix):
        return {k[len(prefix):]: self[k] for k in self.keys() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def assert_close(a, b, rtol=1e-4, atol=1e-6, msg=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{msg} shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(f"{msg} mismatch: {int(bad.sum())}/{a.numel()} elements, worst |a-b|={err.flatten()[i]:.3e} "
                             f"a={a.flatten()[i]:.6e} b={b.flatten()[i]:.6e}")


def rel_err(a, b):
    """||a-b|| / ||b|| (global relative error)."""
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
