"""ctypes binding of libvireo_hip.so (C ABI: include/vireo_hip.h).

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible,
every compute entry point raises.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIREO_LIB", os.path.join(_HERE, "libvireo_hip.so"))  # (override: A/B builds)

KIND_VIREO, KIND_BMM = 0, 1
PROBLEM_BALANCED = 1
STEP_THETA, STEP_GT, STEP_ID, STEP_LOGLIK, STEP_ELBO, STEP_SOFTMAX = 1, 2, 3, 4, 5, 6
KERN_VARIANT_PASS, KERN_CELL_PASS, KERN_DENSE, KERN_COUNT = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128


class VrxError(RuntimeError):
    """An error reported by libvireo_hip.so."""


class ModelCfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_donor", C.c_int32), ("n_gt", C.c_int32),
                ("learn_gt", C.c_int32), ("learn_theta", C.c_int32),
                ("ase_mode", C.c_int32), ("fix_beta_sum", C.c_int32),
                ("n_batch", C.c_int32)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I32 = C.POINTER(C.c_int32)
_I64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); exactly the declarations of include/vireo_hip.h
SIGNATURES = {
    "vrx_last_error": (C.c_char_p, []),
    "vrx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vrx_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), _I64]),
    "vrx_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "vrx_problem_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, _I64, _I32, _I32,
                                     _I32, C.POINTER(_P)]),
    "vrx_problem_create2": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, _I64, _I32, _I32,
                                      _I32, C.c_int32, C.POINTER(_P)]),
    "vrx_problem_build_info": (C.c_int, [_P, _D]),
    "vrx_problem_destroy": (None, [_P]),
    "vrx_problem_binom_const": (C.c_int, [_P, _D]),
    "vrx_problem_n_vars": (C.c_int, [_P, _I32]),
    "vrx_problem_digest": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "vrx_model_create": (C.c_int, [_P, C.POINTER(ModelCfg), C.POINTER(_P)]),
    "vrx_model_destroy": (None, [_P]),
    "vrx_model_set_state": (C.c_int, [_P, _D, _D, _D, _D]),
    "vrx_model_get_state": (C.c_int, [_P, _D, _D, _D, _D]),
    "vrx_model_set_state_raw": (C.c_int, [_P, _D, _D, _D, _D]),
    "vrx_model_stage_reserve": (C.c_int, [_P]),
    "vrx_model_stage_raw": (C.c_int, [_P, C.c_int32, _D, _D]),
    "vrx_model_set_state_staged": (C.c_int, [_P, C.c_int32, _D, _D]),
    "vrx_model_snapshot": (C.c_int, [_P, C.c_int32]),
    "vrx_model_set_restart": (C.c_int, [_P, C.c_int32, _D, _D, _D, _D, C.c_int32]),
    "vrx_model_copy_restart": (C.c_int, [_P, _P, C.c_int32]),
    "vrx_model_set_prior": (C.c_int, [_P, _D, C.c_int64, _D, C.c_int64, _D, _D, C.c_int64]),
    "vrx_model_fit": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, C.c_int32, _D, _I32, _I32]),
    "vrx_model_step": (C.c_int, [_P, C.c_int32, _D]),
    "vrx_model_get_loglik": (C.c_int, [_P, _D]),
    "vrx_model_set_loglik": (C.c_int, [_P, _D]),
    "vrx_model_get_elbo_parts": (C.c_int, [_P, _D]),
    "vrx_problem_donor_reads": (C.c_int, [_P, C.c_int64, _D, _D, _D]),
    "vrx_problem_doublet": (C.c_int, [_P, C.c_int64, C.c_int64, _D, _D, _D, _D, C.c_int64, _D,
                                      C.c_int64, _D, _D]),
    "vrx_problem_cell_loglik": (C.c_int, [_P, C.c_int64, C.c_int64, _D, _D, _D, _D, C.c_int64, _D,
                                          C.c_int64, _D, _D]),
    "vrx_model_info": (C.c_int, [_P, _I32]),
    "vrx_model_profile": (C.c_int, [_P, C.c_int32]),
    "vrx_model_profile_read": (C.c_int, [_P, _D, _I64]),
    "vrx_model_run_iters": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _D]),
    "vrx_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "vrx_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(_P)]),
    "vrx_comm_destroy": (None, [_P]),
    "vrx_comm_allgather_f64": (C.c_int, [_P, _D, C.c_int64, _D]),
    "vrx_comm_barrier": (C.c_int, [_P]),
    "vrx_comm_bcast_f64": (C.c_int, [_P, _D, C.c_int64, C.c_int]),
    "vrx_comm_info": (C.c_int, [_P, _I32]),
    "vrx_comm_bcast_model": (C.c_int, [_P, _P, C.c_int]),
    "vrx_mt19937_random_sample": (C.c_int, [C.POINTER(C.c_uint32), _I32, _D, C.c_int64]),
    "vrx_mt19937_skip": (C.c_int, [C.POINTER(C.c_uint32), _I32, C.c_int64, C.c_int32]),
    "vrx_np_sum_f32": (C.c_int, [C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float)]),
    "vrx_mtx_header": (C.c_int, [C.c_char_p, _I64, _I64, _I64]),
    "vrx_merge_counts": (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                   _P, _P, _P, C.c_int, C.c_int, C.c_int, _I64, _I32, _I32, _I32,
                                   C.c_int]),
    "vrx_mtx_read": (C.c_int, [C.c_char_p, C.c_int64, _I32, _I32, _I32, C.c_int]),
    "vrx_coo_to_csc": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, _I32, _I32, _I32, _I64, _I32, _I64, _I32,
                                 C.c_int]),
    "vrx_mtx_write": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, _I32, _I32, _I32]),
    "vrx_write_table": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _P, _D, C.c_int64, C.c_int64,
                                  C.c_char_p, C.c_int32]),
    "vrx_write_vcf_records": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _P, _P, _P, _P, _P,
                                        C.c_int64, C.c_int64, C.c_int32]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VrxError(
                "libvireo_hip.so is not built (%s). There is no CPU fallback: run "
                "`python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        raise VrxError("libvireo_hip: %s (code %d)" % (lib().vrx_last_error().decode(), rc))


def device_count():
    n = C.c_int(0)
    check(lib().vrx_device_count(C.byref(n)))
    return n.value


def device_info(device=0):
    name = C.create_string_buffer(256)
    cu = C.c_int(0)
    mem = C.c_int64(0)
    check(lib().vrx_device_info(device, name, 256, C.byref(cu), C.byref(mem)))
    return dict(name=name.value.decode(), n_cu=cu.value, hbm_bytes=mem.value)


def device_pci_bus_id(device=0):
    buf = C.create_string_buffer(32)
    check(lib().vrx_device_pci_bus_id(device, buf, 32))
    return buf.value.decode()


def require_gpu():
    n = device_count()
    if n < 1:
        raise VrxError("no HIP device visible: vireo_amd runs on MI355X (gfx950) only and has "
                       "no CPU fallback")
    return n


def dptr(a):
    """double* of a C-contiguous float64 array (or NULL for None)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_D)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
