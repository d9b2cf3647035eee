"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/sedd_hip.h declares, the
ctypes signatures in _lib.py match the header, argument validation works without a GPU (no compute calls), and the
product path refuses to run without the HIP extension / on CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sedd_hip.h")


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|const char\*)\s+(sedd_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        sig = ""
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a or a.startswith("sedd_stream_t"):
                sig += "p"
            elif a.startswith("int64_t") or a.startswith("uint64_t"):
                sig += "l"
            elif a.startswith("float"):
                sig += "f"
            elif a.startswith("int"):
                sig += "i"
            else:
                raise AssertionError(f"unparsed argument '{a}' of {name}")
        protos[name] = (ret, sig)
    return protos


def test_header_symbols_exported_and_signatures_match():
    from protein_lig_sedd_amd import _lib
    L = _lib.lib()
    protos = _prototypes()
    assert len(protos) >= 35
    for name, (ret, sig) in protos.items():
        assert hasattr(L, name), f"libsedd_hip.so does not export {name}"
        if name in ("sedd_abi_version", "sedd_last_error", "sedd_ln_bwd_chunks", "sedd_embedding_bwd_ws_floats",
                    "sedd_gemm_get_cu_budget"):
            continue
        assert name in _lib.SIGNATURES, f"_lib.SIGNATURES lacks {name}"
        assert _lib.SIGNATURES[name] == sig, f"{name}: ctypes '{_lib.SIGNATURES[name]}' vs header '{sig}'"
    for name in _lib.SIGNATURES:
        assert name in protos, f"{name} bound in _lib.py but not declared in the header"
    assert L.sedd_abi_version() == 6


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch, with a message; nothing here needs a device."""
    from protein_lig_sedd_amd import _lib
    L = _lib.lib()
    # unknown graph enum
    assert L.sedd_sample_transition(7, None, None, None, None, None, 0, 1, 26, None) == -1
    assert b"unknown graph" in L.sedd_last_error()
    # M not a multiple of L
    assert L.sedd_score_entropy_fwd(0, None, 0, None, None, None, None, None, None, 10, 4, 26, None) == -1
    # empty problems are fine without pointers
    assert L.sedd_score_entropy_fwd(0, None, 0, None, None, None, None, None, None, 0, 4, 26, None) == 0
    assert L.sedd_predictor_step(0, 0, None, 1, None, None, None, None, 0, 8, 26, None) == 0
    assert L.sedd_predictor_step_philox(0, 1, None, 1, None, None, 1, 2 ** 32, None, 0, 8, 26, None) == -1
    assert b"32 bits" in L.sedd_last_error()
    # unsupported shapes report SEDD_ERR_UNSUPPORTED_SHAPE (-2)
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    p = (p + 15) // 16 * 16
    assert L.sedd_gemm_nt(p, 12, p, 12, p, 4, 1, 1, None, 0, None, 0, 4, 4, 12, None) == -2
    assert b"multiples of 8" in L.sedd_last_error()
    assert L.sedd_attn_fwd(p, p, p, p, p, None, 1, 1, 1, 4, 4, 48, 48, 48, 48, 48, 48, 48, None) == -2
    # fp32 NN / TN products: the k-major operand's extent must be a multiple of 4 floats; fp32 operands give fp32 results
    assert L.sedd_gemm_nn(p, 8, p, 8, p, 8, 0, 0, None, 0, 4, 6, 8, None) == -2
    assert b"gemm_nn (fp32)" in L.sedd_last_error()
    assert L.sedd_gemm_nn_ws(p, 8, p, 8, p, 8, 0, 1, None, 0, 4, 8, 8, None, 0, None) == -2
    assert L.sedd_gemm_nn_ws(p, 8, p, 8, p, 8, 0, 0, None, 0, 4, 8, 8, None, -1, None) == -1
    assert L.sedd_gemm_tn(p, 8, p, 8, p, 8, 0, 0, 0, 6, 8, 5, None, 0, None) == -2
    assert b"gemm_tn (fp32)" in L.sedd_last_error()
    assert L.sedd_gemm_nn(p, 8, p, 8, p, 8, 0, 0, None, 0, 0, 8, 8, None) == 0     # empty problem
    assert L.sedd_adamw_ema_double_step(None, None, None, None, None, None, None, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0,
                                        0.999, 8, None) == -1                       # step is 1-based
    assert L.sedd_adamw_ema_double_step(None, None, None, None, None, None, None, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1,
                                        0.999, 6, None) == -2                       # n % 4
    assert L.sedd_ln_modulate_fwd(p, p, None, None, None, 0, 0, p, 0, None, None, 4, 2, 4100, None) == -2
    assert L.sedd_gemm_get_cu_budget() == 256 and L.sedd_gemm_set_cu_budget(250) == -1 and L.sedd_gemm_set_cu_budget(0) == -1
    assert L.sedd_gemm_set_cu_budget(248) == 0 and L.sedd_gemm_get_cu_budget() == 248 and L.sedd_gemm_set_cu_budget(256) == 0
    assert L.sedd_ln_bwd_chunks(64 * 512, 512) == 64 * 8   # 64-row chunks from 16384 rows up
    assert L.sedd_ln_bwd_chunks(4 * 512, 512) == 4 * 16     # 32-row chunks below
    assert L.sedd_ln_bwd_chunks(10 * 33, 33) == 10 * 2


def test_no_cpu_fallback():
    from protein_lig_sedd_amd import graph_lib, ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gumbel_argmax(torch.rand(4, 26), torch.rand(4, 26))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        graph_lib.Absorbing(25).sample_transition(torch.zeros(2, 4, dtype=torch.int64), torch.ones(2, 1), u=torch.rand(2, 4))


def test_missing_library_fails_loudly(monkeypatch):
    from protein_lig_sedd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsedd_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_host_side_interfaces():
    """registry / factory / error behaviour of the reference interfaces (no tensors touched)."""
    import types
    from protein_lig_sedd_amd import graph_lib, noise_lib, sampling
    ns = types.SimpleNamespace
    g = graph_lib.get_graph(ns(graph=ns(type="absorb"), tokens=25))
    assert g.dim == 26 and g.absorb
    u = graph_lib.get_graph(ns(graph=ns(type="uniform"), tokens=25), tokens=40)
    assert u.dim == 40 and not u.absorb
    with pytest.raises(ValueError):
        graph_lib.get_graph(ns(graph=ns(type="nope"), tokens=3))
    with pytest.raises(ValueError):
        noise_lib.get_noise(ns(noise=ns(type="nope")))
    n = noise_lib.get_noise(ns(noise=ns(type="loglinear", eps=1e-3)))
    s, ds = n(torch.tensor([0.5]))
    assert abs(float(s) - 0.69215) < 1e-4
    assert abs(float(noise_lib.get_noise(ns(noise=ns(type="geometric", sigma_min=1e-4, sigma_max=20.0)))(torch.tensor([1.0]))[0]) - 20.0) < 1e-4
    for name in ("euler", "analytic", "none"):
        assert sampling.get_predictor(name) is not None
    with pytest.raises(ValueError):
        sampling.register_predictor(sampling.EulerPredictor, name="euler")
    assert [graph_lib.Absorbing.curriculum_factor(s) for s in (0, 250, 700, 1200, 1700, 5000)] == \
        [3, 2.5, 2, 2 - 0.8 * 0.8, 1.2, 1]
    assert torch.equal(g.sample_limit(2, 3), torch.full((2, 3), 25))
