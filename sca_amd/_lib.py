"""ctypes binding of libsca_hip.so (include/sca_hip.h).  No fallback: if the library is missing the import of
anything that needs it raises."""
import ctypes as C
import os

import numpy as np

from . import build as _build

K = 16
ACTION_DIM = 7
DIAG_DIM = 5

dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
bp = C.POINTER(C.c_uint8)


class Params(C.Structure):
    _fields_ = [('neighbor_dist', C.c_double), ('time_step', C.c_double), ('time_horizon', C.c_double),
                ('max_speed', C.c_double), ('max_heading_change', C.c_double), ('near_goal_threshold', C.c_double),
                ('max_neighbors', C.c_int32), ('struct_bytes', C.c_int32), ('dt_nominal', C.c_double)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against include/sca_hip.h
SIGNATURES = {
    'sca_default_params': (None, [C.c_void_p]),                   # (version-100 form: 56 bytes)
    'sca_default_params_v2': (None, [C.POINTER(Params), C.c_int32]),
    'sca_version': (C.c_int, []),
    'sca_create': (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'sca_destroy': (None, [C.c_void_p]),
    'sca_last_error': (C.c_char_p, [C.c_void_p]),
    'sca_set_obstacles': (C.c_int, [C.c_void_p, C.c_int, dp, dp]),
    'sca_set_agents': (C.c_int, [C.c_void_p, C.c_int, dp, dp, dp, bp, bp, dp]),
    'sca_set_agent_params': (C.c_int, [C.c_void_p, C.c_int, dp, ip, dp, dp, dp, dp, dp]),
    'sca_set_state': (C.c_int, [C.c_void_p, dp, fp, dp, bp, dp, ip]),
    'sca_get_state': (C.c_int, [C.c_void_p, dp, fp, dp, bp, dp, ip]),
    'sca_set_kd_perm': (C.c_int, [C.c_void_p, ip]),
    'sca_get_kd_perm': (C.c_int, [C.c_void_p, ip]),
    'sca_get_kd_tree': (C.c_int, [C.c_void_p, dp]),
    'sca_set_vpref': (C.c_int, [C.c_void_p, dp, bp]),
    'sca_policy_pass': (C.c_int, [C.c_void_p, C.c_int]),
    'sca_get_actions': (C.c_int, [C.c_void_p, fp]),
    'sca_get_neighbors': (C.c_int, [C.c_void_p, ip, ip, bp, dp, bp]),
    'sca_get_nbr0': (C.c_int, [C.c_void_p, dp]),
    'sca_get_diag': (C.c_int, [C.c_void_p, ip, ip, dp]),
    'sca_env_update': (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    'sca_run_steps': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'sca_env_step': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    'sca_synchronize': (C.c_int, [C.c_void_p]),
    'sca_active_count': (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    'sca_set_shard': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'sca_public_records': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'sca_bind_public_records': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    'sca_step_begin': (C.c_int, [C.c_void_p, C.c_int]),
    'sca_step_end': (C.c_int, [C.c_void_p]),
    'sca_set_stream': (C.c_int, [C.c_void_p, C.c_void_p]),
    'sca_use_own_stream': (C.c_int, [C.c_void_p]),
    'sca_comm_probe': (C.c_int, []),
    'sca_comm_unique_id': (C.c_int, [C.c_void_p]),
    'sca_comm_init': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'sca_comm_destroy': (C.c_int, [C.c_void_p]),
    'sca_partition_init': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.c_int]),
    'sca_partition_disable': (C.c_int, [C.c_void_p]),
    'sca_partition_message_bytes': (C.c_int64, [C.c_void_p]),
    'sca_partition_pack': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'sca_partition_unpack': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'sca_partition_commit': (C.c_int, [C.c_void_p]),
    'sca_partition_counts': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'sca_partition_owned': (C.c_int, [C.c_void_p, ip, C.POINTER(C.c_int)]),
    'sca_last_kernel_ms': (C.c_int, [C.c_void_p, fp, fp, fp]),
    'sca_last_exchange_ms': (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    'sca_last_kd_build_ms': (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    'sca_auto_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    'sca_last_replan_ms': (C.c_int, [C.c_void_p, fp]),
    'sca_last_pass_forms': (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    'sca_set_shard_emulation': (C.c_int, [C.c_void_p, C.c_int]),
    'sca_set_profiling': (C.c_int, [C.c_void_p, C.c_int]),
    'sca_agent_steps': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    'sca_selftest_l3norm': (C.c_int, [C.c_void_p, C.c_int, dp, dp, dp, dp]),
    'sca_selftest_libm': (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp, dp, dp]),
    'sca_selftest_libm_host': (C.c_int, [C.c_int, C.c_int, dp, dp, dp]),
    'sca_history_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'sca_history_rows': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'sca_get_history': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, fp]),
    'sca_tracker_create': (C.c_void_p, [C.c_int, dp, dp, dp, bp, C.c_double, C.c_double, C.c_double, C.c_double]),
    'sca_tracker_set_neighbor_dist': (C.c_int, [C.c_void_p, dp]),
    'sca_tracker_set_agent_params': (C.c_int, [C.c_void_p, dp, dp, dp]),
    'sca_tracker_destroy': (None, [C.c_void_p]),
    'sca_tracker_vpref': (C.c_int, [C.c_void_p, dp, fp, dp, bp, dp, dp, C.c_int]),
    'sca_tracker_replans': (C.c_int, [C.c_void_p, ip]),
    'sca_tracker_debug': (C.c_int, [C.c_void_p, C.c_int, dp]),
    'sca_device_tracker_debug': (C.c_int, [C.c_void_p, C.c_int, dp]),
    'sca_device_tracker_enable': (C.c_int, [C.c_void_p, dp, C.c_double, C.c_double, C.c_double, C.c_int]),
    'sca_device_tracker_disable': (C.c_int, [C.c_void_p]),
    'sca_device_tracker_set_agent_params': (C.c_int, [C.c_void_p, C.c_int, dp, dp, dp]),
    'sca_device_tracker_vpref': (C.c_int, [C.c_void_p, dp, dp]),
    'sca_device_tracker_replans': (C.c_int, [C.c_void_p, ip]),
    'sca_selftest_dubins_words': (C.c_int, [C.c_int, dp, dp, dp, C.POINTER(C.c_int64)]),
    'sca_selftest_plan3d_lean': (C.c_int, [C.c_int, dp, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'sca_libm_check': (C.c_int, [C.POINTER(C.c_int64)]),
    'sca_dubins_plan': (C.c_int, [dp, dp, C.c_double, C.c_double, C.c_double, dp, C.c_char_p, ip, dp, C.c_int]),
    'sca_candidate_table': (C.c_int, [C.c_int, dp, dp]),
    'sca_kd_build_host': (C.c_int, [C.c_int, dp, ip, dp]),
}

_LIB = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's): whichever of the two a
    process loads first is the one everybody gets.  With torch first that is torch's, and libsca_hip.so runs on it (bench.py, the
    multi-GPU steppers); with libsca_hip.so first it would be /opt/rocm's, and a later `import torch` finds no GPU
    (torch.cuda.is_available() == False, measured).  So when a torch installation is present but not imported yet, its runtime is loaded
    here, before the library: either import order then works.  SCA_OWN_HIP_RUNTIME=1 skips this (the system's runtime, no torch later)."""
    import importlib.util
    import sys
    if os.environ.get('SCA_OWN_HIP_RUNTIME') or 'torch' in sys.modules:
        return None
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if not os.path.exists(cand):
        return None
    try:
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def lib():
    """Loads sca_amd/lib/libsca_hip.so.  Raises if it has not been built (python -m sca_amd.build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_build.LIB):
            raise RuntimeError(f'{_build.LIB} is missing: build it with `python -m sca_amd.build` '
                               '(there is no CPU fallback)')
        _share_torch_hip_runtime()
        L = C.CDLL(_build.LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def as_d(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)
