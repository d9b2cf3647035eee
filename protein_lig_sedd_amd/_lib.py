"""ctypes binding of libsedd_hip.so (C ABI declared in include/sedd_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is absent this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEDD_HIP_LIB") or os.path.join(_HERE, "libsedd_hip.so")  # (override: A/B builds)

# enums (include/sedd_hip.h)
F32, BF16 = 0, 1
ABSORB, UNIFORM = 0, 1
STEP_EULER, STEP_ANALYTIC, STEP_DENOISE = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_GRAD = 0, 1, 2, 3

_T = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_int64, "f": ctypes.c_float}

# name -> argument type string (p = pointer / stream, i = int, l = int64, f = float)
SIGNATURES = {
    "sedd_sample_transition": "ipppppllip",
    "sedd_score_entropy_fwd": "ipippppppllip",
    "sedd_score_entropy_bwd": "ipipppppllip",
    "sedd_gumbel_argmax": "pipplip",
    "sedd_predictor_step": "iipippppllip",
    "sedd_predictor_step_philox": "iipippllpllip",
    "sedd_philox_uniform": "llplip",
    "sedd_transp_rate": "ipplip",
    "sedd_reverse_rate": "ippplip",
    "sedd_staggered_score": "ippplli" + "p",
    "sedd_transp_transition": "ippplli" + "p",
    "sedd_timestep_embedding": "ppiip",
    "sedd_cond_combine_fwd": "pppppppp" + "ppppppp" + "p" + "iffiip",
    "sedd_cond_combine_bwd": "pppppppp" + "ppppppp" + "p" + "iiiip",
    "sedd_silu_fwd": "pplp",
    "sedd_silu_bwd": "ppplp",
    "sedd_embedding_fwd": "pppliip",
    "sedd_embedding_bwd": "ppppliip",
    "sedd_ln_modulate_fwd": "pppppilpippllip",
    "sedd_ln_modulate_bwd": "pipppilpppppllip",
    "sedd_ln_modulate_bwd_gate": "pipppilppppp" + "pp" + "flf" + "pp" + "llip",
    "sedd_ln_modulate_bwd_gate_colsum": "pipppilppppp" + "pp" + "flf" + "pp" + "llip",
    "sedd_ln_modulate_bwd_reduce": "ppppl" + "llip",
    "sedd_ln_modulate_bwd_noalloc": "pipppilpppp" + "pppl" + "llip",
    "sedd_gate_residual_fwd": "ppipilpflfpllip",
    "sedd_gate_residual_bwd": "ppipilpflfppillip",
    "sedd_gate_residual_bwd_reduce": "pplp" + "llip",
    "sedd_gate_residual_ln_modulate_fwd": "ppipppil" + "flf" + "ppppp" + "llip",
    "sedd_rotary_qkv": "pippiiiiip",
    "sedd_gelu_fwd": "ppilp",
    "sedd_gelu_bwd": "pppilp",
    "sedd_logits_epilogue": "pippillip",
    "sedd_logits_epilogue_bwd": "piplip",
    "sedd_colsum": "pilppilip",
    "sedd_transpose": "pilpilllp",
    "sedd_cast": "pipilp",
    "sedd_scale_": "piplp",
    "sedd_gemm_nt": "plplpliipipilllp",
    "sedd_gemm_nt_ws": "plplpliipipilllplp",
    "sedd_gemm_tn": "plplpliiilllplp",
    "sedd_gemm_tn_colsum": "plplplpiilllplp",
    "sedd_gemm_nt_dgelu": "plplpliippplllp",
    "sedd_gemm_nn": "plplpliipilllp",
    "sedd_gemm_nn_ws": "plplpliipilllplp",
    "sedd_gemm_nt_rotary": "plplpliipippiiilllp",
    "sedd_gemm_nn_dgelu": "plplpliippplllp",
    "sedd_gemm_set_cu_budget": "i",
    "sedd_gemm_nt_dmul": "plplpliippplllp",
    "sedd_gemm_nn_dmul": "plplpliippplllp",
    "sedd_attn_fwd": "pppppp" + "iiiiii" + "llllll" + "p",
    "sedd_attn_bwd": "ppppppp" + "pppp" + "iiiiii" + "llllll" + "ppp",
    "sedd_attn_varlen_fwd": "pppppp" + "iiiiii" + "lll" + "p",
    "sedd_attn_varlen_bwd": "pppppp" + "pppp" + "p" + "iiiiii" + "lll" + "p",
    "sedd_adamw_ema_step": "ppppppp" + "ffffff" + "ii" + "f" + "lp",
    "sedd_adamw_ema_double_step": "ppppppp" + "ffffff" + "ii" + "f" + "lp",
    "sedd_adamw_ema_step_dev": "pppppppp" + "fffff" + "ii" + "lp",
    "sedd_adamw_step_scalars": "ffffiifp",
    "sedd_write_step_scalars": "ppfffffflp",
    "sedd_sumsq_clip": "plffppp",
    "sedd_clip_from_sumsq": "pffp",
    "sedd_ema_update": "ppflp",
    "sedd_batch_weighted_sum": "ppllppp",
}


class SeddError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libsedd_hip.so (once). Raises ImportError when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C protein_lig_sedd_amd/csrc). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.sedd_abi_version.restype = ctypes.c_int
    L.sedd_last_error.restype = ctypes.c_char_p
    for name, sig in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
        fn.argtypes = [_T[c] for c in sig]
        fn.restype = ctypes.c_int
    L.sedd_gemm_get_cu_budget.argtypes = []
    L.sedd_gemm_get_cu_budget.restype = ctypes.c_int
    L.sedd_ln_bwd_chunks.argtypes = [ctypes.c_int64, ctypes.c_int64]
    L.sedd_ln_bwd_chunks.restype = ctypes.c_int64
    L.sedd_embedding_bwd_ws_floats.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.sedd_embedding_bwd_ws_floats.restype = ctypes.c_int64
    _lib = L
    return L


def check(rc, name):
    if rc != 0:
        raise SeddError(f"{name} failed ({rc}): {lib().sedd_last_error().decode()}")
