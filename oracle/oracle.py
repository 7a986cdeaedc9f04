"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (sca_amd) never does.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
K = 16

POL_SCA, POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP, POL_RVO_DUBINS = 0, 1, 2, 3, 4, 5
FLAG_AT_GOAL, FLAG_COLLISION, FLAG_TIMEOUT = 1, 2, 4


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'sca_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get('SCA_ORACLE_SO') or os.path.join(_HERE, 'liboracle.so')    # SCA_ORACLE_SO: the sanitizer build
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp, fp, ip, bp = (C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8))
        L.orc_round5_py.restype = C.c_double
        L.orc_round5_py.argtypes = [C.c_double]
        L.orc_round5_np.restype = C.c_double
        L.orc_round5_np.argtypes = [C.c_double]
        L.orc_trunc5.restype = C.c_double
        L.orc_trunc5.argtypes = [C.c_double]
        for nm in ('orc_l3norm', 'orc_l3normsq', 'orc_distance'):
            getattr(L, nm).restype = C.c_double
            getattr(L, nm).argtypes = [dp, dp]
        L.orc_l3norm_mixed.restype = C.c_double
        L.orc_l3norm_mixed.argtypes = [dp, fp]
        L.orc_l3norm_f32zero.restype = C.c_double
        L.orc_l3norm_f32zero.argtypes = [fp]
        L.orc_get_phi.restype = C.c_double
        L.orc_get_phi.argtypes = [dp]
        L.orc_pi_2_pi.restype = C.c_double
        L.orc_pi_2_pi.argtypes = [C.c_double]
        L.orc_mod2pi.restype = C.c_double
        L.orc_mod2pi.argtypes = [C.c_double]
        L.orc_is_intersect.restype = C.c_int
        L.orc_is_intersect.argtypes = [dp, dp, C.c_double, dp]
        L.orc_satisfied_constraint.restype = C.c_int
        L.orc_satisfied_constraint.argtypes = [fp, C.c_double, dp]
        L.orc_cartesian2spherical.restype = None
        L.orc_cartesian2spherical.argtypes = [dp, dp, C.c_int, dp]
        L.orc_candidate_table.restype = C.c_int
        L.orc_candidate_table.argtypes = [C.c_double, C.c_int, dp, C.c_int]
        L.orc_kd_build.restype = None
        L.orc_kd_build.argtypes = [C.c_int, dp, ip, dp]
        L.orc_straight_v_pref.restype = None
        L.orc_straight_v_pref.argtypes = [dp, dp, C.c_double, C.c_int, dp]
        L.orc_set_params.restype = None
        L.orc_set_params.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_set_dt_nominal.restype = None
        L.orc_set_dt_nominal.argtypes = [C.c_double]
        L.orc_set_agent_params.restype = None
        L.orc_set_agent_params.argtypes = [C.c_int, dp, ip, dp, dp, dp, dp, dp]
        L.orc_policy_step.restype = C.c_int
        L.orc_policy_step.argtypes = [C.c_int, C.c_int, dp, fp, dp, dp, dp, bp, dp, bp, bp, dp, bp, ip, dp, dp, dp, fp,
                                      ip, ip, bp, dp, bp, dp, ip, ip, C.c_int]
        L.orc_env_update.restype = C.c_int
        L.orc_env_update.argtypes = [C.c_int, C.c_int, dp, fp, dp, dp, bp, dp, fp, dp, dp, ip, dp, dp]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


DEFAULT_PARAMS = dict(neighbor_dist=10.0, max_neighbors=16, time_step=0.1, time_horizon=10.0, max_speed=1.0,
                      max_heading_change=math.pi / 4, near_goal_threshold=0.5, dt_nominal=0.1)


def set_params(**kw):
    """The solver attributes of agent.py:27-41 / config.py:3 for every following call (process-wide); set_params() restores the
    reference's defaults."""
    p = dict(DEFAULT_PARAMS)
    unknown = set(kw) - set(p)
    if unknown:
        raise TypeError(f'unknown oracle parameter(s): {sorted(unknown)}')
    p.update(kw)
    L = lib()
    L.orc_set_params(p['neighbor_dist'], int(p['max_neighbors']), p['time_step'], p['time_horizon'], p['max_speed'],
                     p['max_heading_change'], p['near_goal_threshold'])
    L.orc_set_dt_nominal(p['dt_nominal'])


def set_agent_params(n=0, neighbor_dist=None, max_neighbors=None, time_step=None, time_horizon=None, max_speed=None, max_heading_change=None,
                     dt_nominal=None):
    """Per-agent solver attributes (the reference keeps them on every Agent object): arrays of n, None = the scene's value (set_params).
    set_agent_params() switches them off again."""
    L = lib()
    keep = []

    def arr(a, dt, ct):
        if a is None or n == 0:
            return None
        b = np.ascontiguousarray(a, dt).reshape(n)
        keep.append(b)
        return _p(b, ct)
    L.orc_set_agent_params(int(n), arr(neighbor_dist, np.float64, C.c_double), arr(max_neighbors, np.int32, C.c_int32), arr(time_step, np.float64, C.c_double),
                           arr(time_horizon, np.float64, C.c_double), arr(max_speed, np.float64, C.c_double), arr(max_heading_change, np.float64, C.c_double),
                           arr(dt_nominal, np.float64, C.c_double))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _d(a):
    return _p(a, C.c_double)


def vec3(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def policy_step(pos, vel, heading, radius, pref_speed, flags, goal, policy, zaxis, vpref_ext, vpref_mode, perm,
                obs_pos, obs_radius, nthreads=1):
    """One pass of the first loop of MACAEnv._take_action (mampenv.py:28-40).
    Returns a dict; `flags` and `perm` are copied, the updated copies are returned."""
    L = lib()
    n = int(len(radius))
    m = int(len(obs_radius))
    pos = np.ascontiguousarray(pos, np.float64).reshape(n, 3)
    vel = np.ascontiguousarray(vel, np.float32).reshape(n, 3)
    heading = np.ascontiguousarray(heading, np.float64).reshape(n, 3)
    radius = np.ascontiguousarray(radius, np.float64)
    pref_speed = np.ascontiguousarray(pref_speed, np.float64)
    flags = np.array(flags, np.uint8, copy=True)
    goal = np.ascontiguousarray(goal, np.float64).reshape(n, 3)
    policy = np.ascontiguousarray(policy, np.uint8)
    zaxis = np.ascontiguousarray(zaxis, np.uint8)
    vpref_ext = np.ascontiguousarray(np.nan_to_num(vpref_ext), np.float64).reshape(n, 3)
    vpref_mode = np.ascontiguousarray(vpref_mode, np.uint8)
    perm = np.array(perm, np.int32, copy=True)
    obs_pos = np.ascontiguousarray(obs_pos, np.float64).reshape(m, 3)
    obs_radius = np.ascontiguousarray(obs_radius, np.float64)
    out = dict(action64=np.zeros((n, 7)), action=np.zeros((n, 7), np.float32), nbr_n=np.zeros(n, np.int32),
               nbr_id=np.full((n, K), -1, np.int32), nbr_kind=np.zeros((n, K), np.uint8), nbr_dsq=np.zeros((n, K)),
               nbr_valid=np.zeros(n, np.uint8), vpref=np.zeros((n, 3)), diag=np.zeros((n, 5), np.int32),
               status=np.zeros(n, np.int32))
    L.orc_policy_step(n, m, _d(pos), _p(vel, C.c_float), _d(heading), _d(radius), _d(pref_speed), _p(flags, C.c_uint8),
                      _d(goal), _p(policy, C.c_uint8), _p(zaxis, C.c_uint8), _d(vpref_ext), _p(vpref_mode, C.c_uint8),
                      _p(perm, C.c_int32), _d(obs_pos), _d(obs_radius), _d(out['action64']),
                      _p(out['action'], C.c_float), _p(out['nbr_n'], C.c_int32), _p(out['nbr_id'], C.c_int32),
                      _p(out['nbr_kind'], C.c_uint8), _d(out['nbr_dsq']), _p(out['nbr_valid'], C.c_uint8),
                      _d(out['vpref']), _p(out['diag'], C.c_int32), _p(out['status'], C.c_int32), int(nthreads))
    out['flags'] = flags
    out['perm'] = perm
    return out


def env_update(pos, vel, heading, radius, flags, goal, action, total_dist, max_run_dist, step_num, obs_pos, obs_radius):
    """Second loop of _take_action + is_done (mampenv.py:42-59). Arrays are copied; updated copies returned."""
    L = lib()
    n = int(len(radius))
    m = int(len(obs_radius))
    pos = np.array(pos, np.float64, copy=True).reshape(n, 3)
    vel = np.array(vel, np.float32, copy=True).reshape(n, 3)
    heading = np.array(heading, np.float64, copy=True).reshape(n, 3)
    flags = np.array(flags, np.uint8, copy=True)
    total_dist = np.array(total_dist, np.float64, copy=True)
    step_num = np.array(step_num, np.int32, copy=True)
    radius = np.ascontiguousarray(radius, np.float64)
    goal = np.ascontiguousarray(goal, np.float64).reshape(n, 3)
    action = np.ascontiguousarray(action, np.float32).reshape(n, 7)
    max_run_dist = np.ascontiguousarray(max_run_dist, np.float64)
    obs_pos = np.ascontiguousarray(obs_pos, np.float64).reshape(m, 3)
    obs_radius = np.ascontiguousarray(obs_radius, np.float64)
    done = L.orc_env_update(n, m, _d(pos), _p(vel, C.c_float), _d(heading), _d(radius), _p(flags, C.c_uint8), _d(goal),
                            _p(action, C.c_float), _d(total_dist), _d(max_run_dist), _p(step_num, C.c_int32),
                            _d(obs_pos), _d(obs_radius))
    return dict(pos=pos, vel=vel, heading=heading, flags=flags, total_dist=total_dist, step_num=step_num, done=bool(done))
