"""ctypes binding of the C-ABI library `libar_mi355x.so` (include/ar_mi355x.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C auto_round_amd/csrc` for gfx950 and is the
ONLY compute path of this package: there is no CPU or eager-PyTorch fallback.  Loading fails loudly
(`Mi355xLibraryError`) when the shared object is missing.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AR_MI355X_LIB", os.path.join(_PKG, "lib", "libar_mi355x.so"))  # override: kernel A/B builds
CSRC = os.path.join(_PKG, "csrc")

AR_DT_BF16, AR_DT_F16, AR_DT_F32 = 0, 1, 2
AR_ERR_UNSUPPORTED = -1
ABI_VERSION = 27


class Mi355xLibraryError(RuntimeError):
    """The HIP library is missing, stale, or a kernel launch failed."""


# name -> (restype, argtypes); mirrors include/ar_mi355x.h one to one (checked by tests/test_abi.py)
P, F, I, L = c_void_p, c_float, c_int, c_int64
SIGNATURES = {
    "ar_abi_version": (c_int, []),
    "ar_error_string": (c_char_p, [c_int]),
    "ar_group_minmax": (c_int, [P, P, P, L, I, I, P]),
    "ar_group_absmax": (c_int, [P, P, P, L, I, I, P]),
    "ar_qdq_int_fwd": (c_int, [P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, F, F, F, P]),
    "ar_qdq_int_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, F, F, F, P]),
    "ar_sign_sgd": (c_int, [P, P, L, P, P]),
    "ar_qdq_int_bwd_sgd": (c_int, [P, P, P, P, P, P, P, L, I, I, I, I, I, F, F, F, P, P, I, P, P, P, P, P, P]),
    "ar_mse_workspace_bytes": (c_int64, []),
    "ar_mse_loss_fwd_bwd": (c_int, [P, P, P, P, P, F, L, I, F, P, L, P, P]),
    "ar_qdq_int_act_fwd": (c_int, [P, P, P, L, I, I, I, I, I, F, P]),
    "ar_int_act_bwd": (c_int, [P, P, P, L, I, I, I, I, I, F, P]),
    "ar_search_int_scale": (c_int, [P, P, L, P, I, P, P, L, I, I, I, F, P]),
    "ar_outlier_loss_workspace_bytes": (c_int64, []),
    "ar_outlier_mse_loss_fwd_bwd": (c_int, [P, P, P, P, P, F, L, I, F, P, L, L, P, P]),
    "ar_best_loss_update": (c_int, [P, P, P, c_int32, P, P, P]),
    "ar_iter_begin": (c_int, [P, P, I, P, P, I, I, P, P]),
    "ar_gather_rows": (c_int, [P, P, P, L, L, P]),
    "ar_pack_int": (c_int, [P, P, P, F, L, L, I, I, I, I, I, P, P, P, P]),
    "ar_pack_awq": (c_int, [P, P, P, F, L, L, I, I, I, P, P, P, P]),
    "ar_qdq_fp4_fwd": (c_int, [P, P, P, P, F, P, P, P, P, L, I, I, I, F, F, P]),
    "ar_qdq_fp4_bwd_sgd": (c_int, [P, P, P, P, P, F, P, P, L, I, I, I, F, F, P, P, I, P, P, P, P, P, P]),
    "ar_search_fp4_scale": (c_int, [P, P, P, L, P, P, I, P, L, I, I, I, P]),
    "ar_fp4_act_bwd": (c_int, [P, P, P, P, L, I, I, I, P]),
    "ar_pack_fp4": (c_int, [P, P, P, L, L, I, I, I, P, P, P]),
    "ar_rmsnorm_fwd": (c_int, [P, P, P, P, L, I, F, I, P]),
    "ar_rmsnorm_bwd": (c_int, [P, P, P, P, P, P, L, I, I, P]),
    "ar_layernorm_fwd": (c_int, [P, P, P, P, P, P, L, I, F, I, P]),
    "ar_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, L, I, I, P]),
    "ar_swiglu_fwd": (c_int, [P, L, P, L, L, I, P]),
    "ar_swiglu_bwd": (c_int, [P, P, L, L, L, I, P]),
    "ar_moe_expand": (c_int, [P, P, P, P, L, L, I, P]),
    "ar_moe_combine": (c_int, [P, P, P, P, P, L, L, I, I, P]),
    "ar_moe_rowdot": (c_int, [P, P, P, P, L, L, I, P]),
    "ar_rope_fwd": (c_int, [P, L, P, P, L, P, P, P, L, L, I, I, I, I, P]),
    "ar_headnorm_fwd": (c_int, [P, P, P, P, P, L, L, I, I, I, F, I, P]),
    "ar_headnorm_bwd": (c_int, [P, P, P, P, P, L, L, I, I, I, I, P]),
    "ar_transpose16": (c_int, [P, P, L, L, P]),
    "ar_rope_bwd": (c_int, [P, P, P, P, P, L, P, L, L, L, I, I, I, I, P]),
    "ar_rmsnorm_fwd_exact": (c_int, [P, P, P, P, P, P, L, I, F, F, I, I, P]),
    "ar_rmsnorm_bwd_exact": (c_int, [P, P, P, P, P, P, L, I, I, P]),
    "ar_rope_fwd_exact": (c_int, [P, L, P, L, P, P, L, P, P, L, L, I, I, I, I, P]),
    "ar_rope_bwd_exact": (c_int, [P, L, L, L, P, L, L, L, P, P, L, P, L, P, L, L, L, I, I, I, I, P]),
    "ar_swiglu_fwd_exact": (c_int, [P, L, P, L, P, L, L, I, P]),
    "ar_swiglu_bwd_exact": (c_int, [P, P, L, P, L, P, L, P, L, L, L, I, I, P]),
    "ar_layernorm_fwd_exact": (c_int, [P, P, P, P, P, P, L, I, F, I, I, P]),
    "ar_layernorm_bwd_exact": (c_int, [P, P, P, P, P, P, P, L, I, I, P]),
    "ar_gemm_dw": (c_int, [P, P, P, L, L, L, L, L, L, I, P, L, P]),
    "ar_gemm_dw_grouped": (c_int, [P, P, P, L, L, L, L, L, P, P, I, P]),
    "ar_gemm_nt": (c_int, [P, P, P, L, L, L, L, L, L, P]),
    "ar_gemm_nt_grouped": (c_int, [P, P, P, L, L, L, L, L, L, P, P, I, P]),
    "ar_gemm_nt_config": (c_int, [I]),
    "ar_gemm_nt_trace": (c_int, [P, P, P, L, L, L, L, L, L, P, I, P]),
    "ar_gemm_dw_ex": (c_int, [P, P, P, L, L, L, L, L, L, I, P, L, I, P]),
    "ar_gemm_dw_sk": (c_int, [P, P, P, L, L, L, L, L, L, P, L, P, P]),
    "ar_gemm_dw_workspace_bytes": (c_int64, [L, L, L]),
    "ar_gemm_dw_config": (c_int, [I, I]),
    "ar_attn_fwd": (c_int, [P, P, P, P, P, L, L, L, L, F, I, L, L, P]),
    "ar_attn_fwd_masked": (c_int, [P, P, P, P, P, L, L, L, L, F, F, F, L, L, L, P]),
    "ar_attn_fwd_exact": (c_int, [P, P, P, P, P, L, L, L, L, L, F, F, F, L, L, L, L, L, L, L, L, L, L, L, P]),
    "ar_attn_exact_config": (c_int, [I]),
    "ar_attn_bwd_exact_workspace_bytes": (c_int64, [L, L, L]),
    "ar_attn_bwd_exact": (c_int, [P, P, P, P, P, P, P, P, P, L, L, L, L, L, F, F, F, L] + [L] * 18 + [P, L, P]),
    "ar_attn_bwd_workspace_bytes": (c_int64, [L, L, L]),
    "ar_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, L, L, L, L, F, I, L, L, L, L, L, L, L, L, P, L, P]),
    "ar_attn_bwd_masked": (c_int, [P, P, P, P, P, P, P, P, P, L, L, L, L, F, F, F, L, L, L, L, L, L, L, L, L, P, L, P]),
    "ar_profile_enable": (c_int, [I]),
    "ar_profile_reset": (c_int, []),
    "ar_profile_read": (c_int, [I, L, P, P, P]),
}

_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU). Returns the .so path."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise Mi355xLibraryError(f"hipcc build failed (see output above): {' '.join(cmd)}")
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load (once) and type the C-ABI library.  Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Mi355xLibraryError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {CSRC}`); this package has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise Mi355xLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ar_abi_version() != ABI_VERSION:
        raise Mi355xLibraryError(f"ABI mismatch: library {lib.ar_abi_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().ar_error_string(code)
        raise Mi355xLibraryError(f"{what} failed: [{code}] {msg.decode() if msg else '?'}")
