"""ctypes binding of libplonk_hip.so (the C ABI declared in include/plonk_hip.h).

There is no CPU fallback: importing succeeds without the library so that host-side code
and CPU tests can run, but any call that needs it raises `LibraryMissing`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
#: PLONK_HIP_LIB names another build of the same library (kernel A/B measurements)
LIB_PATH = os.environ.get("PLONK_HIP_LIB") or os.path.join(_HERE, "libplonk_hip.so")

PLK_OK = 0
PLK_ERR_LENGTH = -1
PLK_ERR_NOT_POW2 = -2
PLK_ERR_TWO_ADICITY = -3
PLK_ERR_ZERO_DENOMINATOR = -4
PLK_ERR_SIGMA_RANGE = -5
PLK_ERR_HIP = -6
PLK_ERR_ARG = -7
PLK_ERR_NOMEM = -8

NUM_WIRES = 4
NUM_SELECTORS = 9
MAX_BATCH = 24
INPUT_FIELD = 0
INPUT_U64 = 1
NO_WIRE = (1 << 64) - 1
NOALLOC_MAX_ROWS = 1 << 20

u64p = C.POINTER(C.c_uint64)
u64pp = C.POINTER(u64p)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
BUILDER_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, u64p, C.c_size_t)

FIELD_BLS12_381_FR = 0
FIELD_TWEEDLE_FR = 1
FIELD_IDS = {"bls12_381_fr": FIELD_BLS12_381_FR, "tweedle_fr": FIELD_TWEEDLE_FR}

NODE_RCCL = 0
NODE_COPY = 1


class CircuitIO(C.Structure):
    """plk_circuit_io: the columns and outputs of one circuit of plk_node_round23_batch / plk_node_round23 (device addresses)."""
    _fields_ = [("wires", C.c_void_p * NUM_WIRES), ("sel", C.c_void_p * NUM_SELECTORS), ("sigma", C.c_void_p * NUM_WIRES),
                ("beta", C.c_uint64 * 4), ("gamma", C.c_uint64 * 4),
                ("z_out", C.c_void_p), ("gate_out", C.c_void_p), ("perm_out", C.c_void_p)]


#: every symbol include/plonk_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "plk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "plk_ctx_create_field": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "plk_ctx_field": (C.c_int, [C.c_void_p]),
    "plk_field_name": (C.c_char_p, [C.c_int]),
    "plk_field_constants_of": (C.c_int, [C.c_int, u64p, u64p, u64p]),
    "plk_ctx_destroy": (None, [C.c_void_p]),
    "plk_strerror": (C.c_char_p, [C.c_int]),
    "plk_last_error": (C.c_char_p, [C.c_void_p]),
    "plk_last_error_row": (C.c_longlong, [C.c_void_p]),
    "plk_field_constants": (None, [u64p, u64p, u64p]),
    "plk_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "plk_host_free": (None, [C.c_void_p, C.c_void_p]),
    "plk_intt": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p]),
    "plk_build_and_convert": (C.c_int, [C.c_void_p, C.c_size_t, BUILDER_FN, C.c_void_p, u64p]),
    "plk_ntt": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p]),
    "plk_gate_polys": (C.c_int, [C.c_void_p, u64pp, u64pp, C.c_size_t, C.c_size_t, u64pp, u64pp]),
    "plk_gate_constraints": (C.c_int, [C.c_void_p, u64pp, u64pp, C.c_size_t, C.c_size_t, u64p]),
    "plk_grand_product": (C.c_int, [C.c_void_p, u64pp, u64pp, C.c_size_t, u64p, u64p, u64p]),
    "plk_perm_polys": (C.c_int, [C.c_void_p, u64pp, u64pp, C.c_size_t, C.c_size_t, u64p, u64p, u64pp, u64p, u64p]),
    "plk_perm_constraints": (C.c_int, [C.c_void_p, u64p, u64pp, u64pp, C.c_size_t, C.c_size_t, u64p, u64p, C.c_size_t, u64p]),
    "plk_round23": (C.c_int, [C.c_void_p, u64pp, u64pp, u64pp, C.c_size_t, C.c_size_t, u64p, u64p, C.c_size_t, u64p, u64p, u64p]),
    "plk_gate_table_fill": (C.c_int, [C.c_void_p, u8p, u64pp, C.c_size_t, u64p, C.c_size_t, u64p, u64pp, u64pp]),
    "plk_dev_gate_table_fill": (C.c_int, [C.c_void_p, C.c_void_p, u8p, u64pp, C.c_size_t, u64p, C.c_size_t, u64p, u64pp, u64pp]),
    "plk_msm_set_bases": (C.c_int, [C.c_void_p, u64p, C.c_size_t]),
    "plk_msm_g1": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p]),
    "plk_msm_g1_batch": (C.c_int, [C.c_void_p, u64pp, C.c_size_t, C.c_size_t, u64p]),
    "plk_g1_sum": (C.c_int, [u64p, C.c_size_t, u64p]),
    "plk_dev_msm_set_bases": (C.c_int, [C.c_void_p, u64p, C.c_size_t]),
    "plk_dev_msm_g1": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, u64p]),
    "plk_dev_msm_g1_batch": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, C.c_size_t, C.c_size_t, u64p]),
    "plk_dev_g1_multiples": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_size_t, u64p]),
    "plk_coset_ntt": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p, C.c_int, u64p]),
    "plk_quotient": (C.c_int, [C.c_void_p, u64pp, C.c_size_t, u64p, C.c_size_t, C.c_size_t, u64p, u64p]),
    "plk_dev_coset_ntt": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, C.c_size_t, C.c_size_t, C.c_size_t, u64p, C.c_int, u64pp]),
    "plk_dev_quotient": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, C.c_size_t, u64p, C.c_size_t, C.c_size_t, u64p, u64p]),
    "plk_sigma_table": (C.c_int, [C.c_void_p, u64pp, C.c_size_t, u64pp]),
    "plk_dev_sigma_table": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, C.c_size_t, u64pp]),
    "plk_dev_ntt_batch": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, u32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, u64pp]),
    "plk_dev_extend4": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, u32p, C.c_size_t, C.c_size_t, u64pp, u64pp, C.c_size_t, C.c_size_t, C.c_size_t]),
    "plk_dev_ntt_strided": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, u64p, C.c_size_t]),
    "plk_dev_fourstep_cols": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, u64p]),
    "plk_dev_fourstep_rows": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u64p]),
    "plk_dev_mul": (C.c_int, [C.c_void_p, C.c_void_p, u64p, u64p, C.c_size_t, u64p]),
    "plk_dev_twiddle_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, u64p]),
    "plk_dev_grand_product": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, u64pp, C.c_size_t, u64p, u64p, u64p]),
    "plk_dev_grand_product_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, u64pp, C.c_size_t, u64p, u64p, u64p]),
    "plk_dev_grand_product_local": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, u64pp, C.c_size_t, C.c_size_t, C.c_size_t, u64p, u64p, u64p]),
    "plk_dev_grand_product_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, u64p, C.c_size_t, C.c_size_t, u64p]),
    "plk_dev_gate_constraint_eval": (C.c_int, [C.c_void_p, C.c_void_p, u64pp, u64pp, C.c_size_t, u64p]),
    "plk_dev_perm_constraint_eval": (C.c_int, [C.c_void_p, C.c_void_p, u64p, u64pp, u64pp, C.c_size_t, u64p, u64p, C.c_size_t, u64p]),
    "plk_dev_perm_constraint_eval_range": (C.c_int, [C.c_void_p, C.c_void_p, u64p, u64pp, u64pp, C.c_size_t, C.c_size_t, u64p, u64p, C.c_size_t, u64p]),
    # (every pointer as a plain address: the call is made thousands of times a second for small circuits, and a
    # typed ctypes pointer per argument would cost more than the call)
    "plk_dev_round23": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plk_dev_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "plk_profile_begin": (None, [C.c_void_p]),
    "plk_profile_end": (C.c_char_p, [C.c_void_p]),
    "plk_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "plk_dev_free": (None, [C.c_void_p, C.c_void_p]),
    "plk_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plk_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plk_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "plk_node_create": (C.c_int, [C.POINTER(C.c_int), C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "plk_node_create_field": (C.c_int, [C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "plk_node_destroy": (None, [C.c_void_p]),
    "plk_node_ranks": (C.c_size_t, [C.c_void_p]),
    "plk_node_ctx": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "plk_node_last_error": (C.c_char_p, [C.c_void_p]),
    "plk_node_ntt": (C.c_int, [C.c_void_p, u64pp, C.c_size_t, C.c_int, C.c_int, u64pp]),
    "plk_node_ntt_host": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_int, u64p]),
    "plk_node_sync": (C.c_int, [C.c_void_p]),
    "plk_node_round23_batch": (C.c_int, [C.c_void_p, C.POINTER(CircuitIO), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "plk_node_round23": (C.c_int, [C.c_void_p, C.POINTER(CircuitIO), C.c_size_t, C.c_size_t, C.c_size_t]),
    "plk_node_round23_host": (C.c_int, [C.c_void_p, u64pp, u64pp, u64pp, C.c_size_t, C.c_size_t, u64p, u64p, C.c_size_t, u64p, u64p, u64p]),
    "plk_dev_to_montgomery": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, u64p]),
    "plk_dev_from_montgomery": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, u64p]),
}


class LibraryMissing(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def _preload_hip_runtime() -> None:
    """A PyTorch ROCm wheel bundles its own HIP runtime (torch/lib/libamdhip64.so, same soname as
    the system one).  Two HIP runtimes in one process cannot both open the GPU, so when torch is
    installed its copy is loaded first and libplonk_hip.so binds to it by soname; a later
    `import torch` then finds that same file already mapped.  Without torch the system runtime
    on the library's RUNPATH is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libplonk_hip.so and type every entry point.  Raises LibraryMissing if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C plonk_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
