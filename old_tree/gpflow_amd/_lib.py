"""ctypes binding of libgpk.so (include/gpk.h) -- the only door from the Python host into device code.

There is deliberately NO fallback: if the shared library is missing, or a tensor is not an fp64
tensor on a HIP device, the call raises.  (The NumPy oracle under ``oracle/`` is test infrastructure
and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPK_LIBRARY: an explicit path to the shared library -- used by the same-box A/B tooling (tools/ab*.sh) to load the
# experimental build (libgpk_exp.so, environment tunables); unset, the product library next to this file is loaded.
_LIB_PATH = os.environ.get("GPK_LIBRARY") or os.path.join(_HERE, "libgpk.so")

c_void_p, c_int, c_long, c_double, c_size_t = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_size_t
_dp = C.c_void_p  # device pointers travel as integers


class GpkError(RuntimeError):
    pass


_SIGS = {
    "gpk_version": (C.c_char_p, []),
    "gpk_kernel_matrix": (c_int, [c_void_p, c_int, _dp, c_int, c_long, _dp, c_int, c_long, c_int,
                                  C.POINTER(c_double), c_int, c_double, c_double, c_int, _dp, c_long]),
    "gpk_kernel_matrix_hadamard": (c_int, [c_void_p, c_int, _dp, c_int, c_long, _dp, c_int, c_long, c_int,
                                           C.POINTER(c_double), c_int, c_double, _dp, c_long, _dp, c_long]),
    "gpk_kernel_matrix_combine": (c_int, [c_void_p, c_int, c_int, _dp, c_int, c_long, _dp, c_int, c_long, c_int,
                                          C.POINTER(c_double), c_int, c_double, c_double, _dp, c_long, _dp, c_long]),
    "gpk_diag_add": (c_int, [c_void_p, _dp, c_int, c_long, _dp]),
    "gpk_invd_elems": (c_size_t, [c_int, c_int]),
    "gpk_potrf": (c_int, [c_void_p, _dp, c_int, c_int, c_long, c_int, c_long, _dp, c_int, _dp]),
    "gpk_combine_parts": (c_int, [c_void_p, _dp, c_int, c_long, c_int, c_int, c_long, c_double, c_int, c_double, _dp, c_long]),
    "gpk_stream_selfcheck": (c_int, [C.POINTER(c_double), C.POINTER(c_double), C.POINTER(c_int)]),
    "gpk_chain_handoff_mode": (c_int, []),
    "gpk_potrf_inv": (c_int, [c_void_p, _dp, c_int, c_int, c_long, _dp, c_int, _dp]),
    "gpk_trtri_blocks": (c_int, [c_void_p, _dp, c_int, c_long, c_int, c_long, _dp]),
    "gpk_trsm": (c_int, [c_void_p, c_int, _dp, c_long, _dp, c_int, _dp, c_int, c_long, c_int, c_long,
                         c_long]),
    "gpk_transpose_factor": (c_int, [c_void_p, _dp, c_long, _dp, c_int, _dp, c_long, _dp]),
    "gpk_gemm_nt": (c_int, [c_void_p, c_int, c_int, c_int, c_double, _dp, c_long, _dp, c_long, c_double,
                            _dp, c_long, c_int, c_int, c_int, c_long, c_long, c_long]),
    "gpk_transpose": (c_int, [c_void_p, _dp, c_int, c_int, c_long, _dp, c_long, c_int, c_int, c_long,
                              c_long]),
    "gpk_row_stats": (c_int, [c_void_p, _dp, c_int, c_int, c_long, _dp, _dp, c_int, c_double, c_double,
                              _dp, _dp, _dp]),
    "gpk_row_dot": (c_int, [c_void_p, _dp, c_long, _dp, c_long, c_int, c_int, c_double, c_double, _dp]),
    "gpk_row_sumsq": (c_int, [c_void_p, _dp, c_int, c_int, c_long, c_double, c_double, _dp]),
    "gpk_project_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gpk_project": (c_int, [c_void_p, _dp, c_int, c_int, c_long, _dp, c_long, c_int, _dp, _dp, c_size_t]),
    "gpk_project_batched": (c_int, [c_void_p, _dp, c_int, c_int, c_long, c_long, _dp, c_long, c_int, _dp, _dp, c_size_t]),
    "gpk_reduce_workspace_bytes": (c_size_t, [c_int]),
    "gpk_gaussian_varexp_sum": (c_int, [c_void_p, _dp, c_long, _dp, c_int, c_int, _dp, c_int, _dp,
                                        C.POINTER(c_double), c_int, c_double, _dp, c_double, _dp, _dp, _dp,
                                        c_size_t]),
    "gpk_gauss_kl_white": (c_int, [c_void_p, _dp, _dp, c_int, c_int, c_int, _dp, _dp, c_size_t]),
    "gpk_sum_log_diag": (c_int, [c_void_p, _dp, c_int, c_long, c_int, c_long, _dp]),
    "gpk_sumsq": (c_int, [c_void_p, _dp, c_int, c_int, c_long, c_int, _dp, _dp, c_size_t]),
    "gpk_gpr_lml_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gpk_gpr_lml": (c_int, [c_void_p, c_int, _dp, c_int, c_int, c_long, _dp, c_int, c_long,
                            C.POINTER(c_double), c_int, c_double, c_double, _dp, c_double, _dp, _dp, _dp,
                            c_size_t]),
    "gpk_svgp_elbo_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "gpk_svgp_elbo_shard": (c_int, [c_void_p, c_int, _dp, c_int, c_long, _dp, _dp, c_int, c_long, c_long,
                                    c_int, c_int, C.POINTER(c_double), c_int, c_double, c_double, _dp,
                                    c_double, c_double, _dp, _dp, c_int, c_int, _dp, _dp, _dp, c_size_t]),
    "gpk_svgp_elbo_sep_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gpk_svgp_elbo_shard_sep": (c_int, [c_void_p, C.POINTER(c_int), _dp, c_int, c_long, c_long, _dp, _dp, c_int, c_long, c_long,
                                        c_int, c_int, C.POINTER(c_double), c_int, C.POINTER(c_double), c_double, _dp, c_double,
                                        c_double, _dp, _dp, _dp, _dp, _dp, c_size_t]),
    "gpk_moment_rows": (c_int, [c_void_p, _dp, c_long, c_int, c_int, _dp, c_long]),
    "gpk_stationary_adjoint_tail": (c_int, [c_void_p, _dp, c_long, _dp, c_long, c_int, c_int, _dp, c_double, c_int, _dp, _dp, c_long,
                                            _dp, c_int, c_double]),
    "gpk_adam_step": (c_int, [c_void_p, _dp, _dp, _dp, _dp, c_long, c_double, c_double, c_double, c_double, c_int]),
    "gpk_lowrank_axpy": (c_int, [c_void_p, c_double, _dp, c_long, _dp, c_long, _dp, c_long, c_int, c_int, c_int, _dp, c_long]),
    "gpk_symmetrize": (c_int, [c_void_p, _dp, c_int, c_long]),
    "gpk_publish_host": (c_int, [c_void_p, _dp, c_int, _dp, c_void_p, c_int]),
    "gpk_profile_gemm_enable": (None, [c_int]),
    "gpk_profile_gemm_collect": (c_int, [C.POINTER(c_double), C.POINTER(c_long), C.POINTER(c_double)]),
    "gpk_profile_gemm_collect_min": (c_int, [c_double, c_int, C.POINTER(c_double), C.POINTER(c_long), C.POINTER(c_double)]),
    "gpk_profile_gemm_collect_kind": (c_int, [c_int, c_double, C.POINTER(c_double), C.POINTER(c_long), C.POINTER(c_double)]),
    "gpk_profile_gemm_window": (c_int, [c_double, C.POINTER(c_double), C.POINTER(c_double), C.POINTER(c_double),
                                        C.POINTER(c_long)]),
    "gpk_bench_mfma_f64": (c_int, [c_void_p, c_int, c_int, _dp]),
    "gpk_bench_stream_store": (c_int, [c_void_p, _dp, c_long]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGS))

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libgpk.so (once).  Raises ImportError with the build recipe if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gpflow_amd/csrc`). "
            "gpflow_amd has no CPU fallback."
        )
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        names = {-1: "GPK_E_ARG (bad argument)", -2: "GPK_E_WORKSPACE (workspace too small)",
                 -3: "GPK_E_UNSUPPORTED"}
        raise GpkError(f"{what}: {names.get(rc, rc)}")
    raise GpkError(f"{what}: HIP runtime error {rc}")


def host_doubles(values):
    arr = (c_double * len(values))(*[float(v) for v in values])
    return arr
