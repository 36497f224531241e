"""ctypes binding of librtd.so (include/rtd.h).  There is no CPU fallback: if the HIP library is
missing or fails to load, importing the solver entry points raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTD_LIB", os.path.join(_HERE, "librtd.so"))  # RTD_LIB: A/B builds of the same ABI


class rtd_dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("ncols", "nlayers", "nquad", "nleg", "nfourier", "nscoeffs", "nbdrf", "beam")]


_dp = C.c_void_p  # double* arguments are passed as plain addresses (an int from the array interface: 3 x cheaper to form than a
#                    ctypes pointer object, and a one-column pydisort() call hands over twenty of them)
_vp = C.c_void_p


class rtd_inputs(C.Structure):
    """Prepared arguments of one batch for the one-call entry points (include/rtd.h: rtd_inputs)."""
    _fields_ = [(n, _dp) for n in
                ("mu_pos", "weights", "scaled_omega", "tau", "scaled_tau_with_0", "scale_tau", "wleg", "mu0", "I0",
                 "phi0", "rescale", "b_pos", "b_neg", "s_poly", "bdrf_q", "bdrf_q0")]


# every symbol include/rtd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rtd_version": (C.c_int, []),
    "rtd_last_error": (C.c_char_p, []),
    "rtd_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "rtd_device_memory": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rtd_plan_create": (C.c_int, [C.POINTER(rtd_dims), C.c_int32, C.POINTER(_vp)]),
    "rtd_plan_create_windowed": (C.c_int, [C.POINTER(rtd_dims), C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "rtd_plan_create_retained": (C.c_int, [C.POINTER(rtd_dims), C.c_int32, C.c_int32, C.c_int64, C.POINTER(_vp)]),
    "rtd_plan_create_retained_form": (C.c_int, [C.POINTER(rtd_dims), C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(_vp)]),
    "rtd_plan_retained": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "rtd_plan_windows": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtd_plan_destroy": (C.c_int, [_vp]),
    "rtd_plan_synchronize": (C.c_int, [_vp]),
    "rtd_plan_device_bytes": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rtd_pool_set_limit": (C.c_int, [C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "rtd_pool_bytes": (C.c_int, [C.POINTER(C.c_int64)]),
    "rtd_pool_trim": (C.c_int, [C.c_int32, C.POINTER(C.c_int64)]),
    "rtd_plan_get_column_status": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "rtd_plan_set_quadrature": (C.c_int, [_vp, _dp, _dp]),
    "rtd_plan_set_columns": (C.c_int, [_vp] + [_dp] * 14),
    "rtd_plan_set_columns_raw": (C.c_int, [_vp, _dp, _dp, _dp, C.c_int32] + [_dp] * 9),
    "rtd_plan_set_bdrf_samples": (C.c_int, [_vp, C.c_int32, _dp, _dp]),
    "rtd_plan_set_mode_shard": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32]),
    "rtd_plan_invalidate_tables": (C.c_int, [_vp]),
    "rtd_plan_solve": (C.c_int, [_vp]),
    "rtd_solve_batch": (C.c_int, [C.POINTER(rtd_dims), C.c_int32, C.POINTER(rtd_inputs), C.c_int32, _dp, C.c_int32, _dp]
                        + [_dp] * 5),
    "rtd_solve_tensors": (C.c_int, [C.POINTER(rtd_dims), C.c_int32, C.POINTER(rtd_inputs), C.c_int32] + [_dp] * 5),
    "rtd_plan_evaluate": (C.c_int, [_vp, C.c_int32, _dp, C.c_int32, _dp, C.c_int32] + [_dp] * 6),
    "rtd_plan_set_nt": (C.c_int, [_vp, C.c_int32, _dp, _dp, _dp, _dp]),
    "rtd_plan_set_eval_points": (C.c_int, [_vp, C.c_int32, _dp, C.c_int32, _dp]),
    "rtd_plan_run": (C.c_int, [_vp]),
    "rtd_plan_fetch": (C.c_int, [_vp] + [_dp] * 5),
    "rtd_plan_run_fetch": (C.c_int, [_vp] + [_dp] * 5),
    "rtd_plan_result_dev_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(_vp),
                                           C.POINTER(C.c_int64)]),
    "rtd_plan_get_tensors": (C.c_int, [_vp, C.c_int32] + [_dp] * 5),
    "rtd_plan_enable_timing": (C.c_int, [_vp, C.c_int32]),
    "rtd_plan_get_timing": (C.c_int, [_vp, _dp, C.POINTER(C.c_int64), C.c_int32]),
    "rtd_plan_max_sweeps": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "rtd_plan_pivoted_chains": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "rtd_comm_preload": (C.c_int, []),
    "rtd_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rtd_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int32, C.c_int32]),
    "rtd_comm_size": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtd_comm_transport": (C.c_int, [_vp, C.c_char_p, C.c_int32]),
    "rtd_comm_allgather_fluxes": (C.c_int, [_vp]),
    "rtd_comm_allreduce_results": (C.c_int, [_vp]),
    "rtd_plan_solve_layers": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "rtd_comm_allgather_layers": (C.c_int, [_vp, C.c_int32]),
    "rtd_plan_solve_bc": (C.c_int, [_vp]),
    "rtd_comm_allgather_results": (C.c_int, [_vp]),
    "rtd_comm_gather_results": (C.c_int, [_vp, C.c_int32]),
    "rtd_comm_fetch_gathered_results": (C.c_int, [_vp, _dp, _dp]),
    "rtd_comm_fetch_gathered_columns": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "rtd_comm_fetch_gathered": (C.c_int, [_vp, _dp]),
    "rtd_comm_destroy": (C.c_int, [_vp]),
}
RTD_ERR_TAU_RANGE = 3
RTD_ERR_NUMERIC = 5


class NumericalError(RuntimeError, np.linalg.LinAlgError):
    """Numerical failure reported by the device (RTD_ERR_NUMERIC): singular boundary-condition system, non-positive
    Cholesky pivot, unconverged eigen-iteration or 1/mu0 on an eigenvalue.  The reference raises
    ``numpy.linalg.LinAlgError`` from ``np.linalg.solve`` / ``eig`` in these situations
    (_solve_for_gen_and_part_sols.py:179-183, :226-231; _solve_for_coeffs.py:326-333, :383)."""


_lib = None


def load():
    """Load librtd.so and declare every prototype; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python pythonic-disort_amd/build.py` "
                "(pydisort_amd has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def dptr(a):
    """float64 C-contiguous array -> double* (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.__array_interface__["data"][0]


def check(rc):
    if rc != 0:
        msg = load().rtd_last_error().decode()
        if rc == RTD_ERR_TAU_RANGE:
            raise ValueError("tau input outside the tau range specified for the atmosphere (check `tau_arr`).")
        if rc == RTD_ERR_NUMERIC:
            raise NumericalError(msg)
        raise RuntimeError(f"librtd error {rc}: {msg}")
