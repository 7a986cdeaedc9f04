"""The arithmetic the HIP kernel uses (sca_amd/csrc/sca_core.h), compiled for the host by g++ in a
test-only harness, against the golden vectors recorded from the reference.  Bit-exact float32 actions.

This is how the algebraic cone test (no asin/acos), the exact round5, the host-libm posture threshold,
the ORCA planes and LP1-4 are validated on a machine without a GPU.  The harness is not part of the
product (sca_amd never loads it)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from golden_util import episode_fixtures, fixture_agent_params, fixture_params, load, static_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def harness():
    ubsan = bool(os.environ.get('SCA_HARNESS_UBSAN'))          # tests/test_sanitizers.py re-runs this module with it set
    out = os.path.join(ROOT, 'tests', '_build', 'libcore_harness_ubsan.so' if ubsan else 'libcore_harness.so')
    src = os.path.join(ROOT, 'tests', 'core_harness.cpp')
    hdrs = [os.path.join(ROOT, 'sca_amd', 'csrc', h) for h in ('sca_core.h', 'sca_glibc_math.h', 'sca_glibc_tables.h')]    # (sca_core.h includes them)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + hdrs):
        san = ['-O1', '-g', '-fsanitize=undefined', '-fno-sanitize-recover=all'] if ubsan else ['-O2']
        subprocess.check_call(['g++', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-mfma',
                               '-fno-builtin-pow', '-I' + os.path.join(ROOT, 'sca_amd', 'csrc'), '-o', out, src] + san)
    H = C.CDLL(out)
    dp, fp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    H.core_round5_py.restype = C.c_double
    H.core_round5_py.argtypes = [C.c_double]
    H.core_trunc5.restype = C.c_double
    H.core_trunc5.argtypes = [C.c_double]
    for nm in ('core_l3norm', 'core_l3normsq', 'core_distance'):
        getattr(H, nm).restype = C.c_double
        getattr(H, nm).argtypes = [dp, dp]
    H.core_get_phi.restype = C.c_double
    H.core_get_phi.argtypes = [dp]
    H.core_pi_2_pi.restype = C.c_double
    H.core_pi_2_pi.argtypes = [C.c_double]
    H.core_posture_ok.restype = C.c_int
    H.core_posture_ok.argtypes = [C.c_double, C.c_double, fp, C.c_double, dp]
    H.core_is_intersect.restype = C.c_int
    H.core_is_intersect.argtypes = [dp, dp, C.c_double, dp]
    H.core_c2s.restype = None
    H.core_c2s.argtypes = [dp, dp, C.c_int, dp]
    H.core_solve_agent.restype = C.c_int
    H.core_solve_agent.argtypes = [dp, C.c_int, C.c_int, C.c_double, dp, fp, C.c_double, dp, dp, C.c_int, dp, C.c_int,
                                   dp, fp, dp, bp, bp, dp, dp, dp, dp, fp, dp, ip]
    return H


def cos_threshold(mhc):
    lo, hi = -1.0, 1.0
    while np.nextafter(lo, 2.0) < hi:
        mid = lo + (hi - lo) * 0.5
        if mid <= lo or mid >= hi:
            break
        if math.acos(mid) <= mhc:
            hi = mid
        else:
            lo = mid
    return hi


@pytest.fixture(scope='module')
def tables():
    from sca_amd import _lib
    L = _lib.lib()
    out = {}
    for n in (256, 128):
        unit = np.zeros(3 * n)
        phi = np.zeros(n)
        assert L.sca_candidate_table(n, _lib.ptr(unit, C.c_double), _lib.ptr(phi, C.c_double)) == 0
        out[n] = (unit, phi)
    return out


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_round5_exact(harness):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, 500, 200000), rng.uniform(0, 3, 100000),
                         np.arange(0, 4000) * 1e-5 + 5e-6, [0.285, 1.005, 2.675, 0.000005, 1e-5, 0.0]])
    for x in xs:
        assert harness.core_round5_py(float(x)) == round(float(x), 5), x


def test_scalar_helpers_vs_golden(harness):
    fx = load('F8_kat')
    n = len(fx['a'])
    a = np.ascontiguousarray(fx['a']); b = np.ascontiguousarray(fx['b'])
    # x*x stands in for pow(x,2): equality is still expected on these vectors (no value sits on a rounding edge)
    assert all(harness.core_l3norm(_d(a[i]), _d(b[i])) == fx['l3'][i] for i in range(n))
    assert all(harness.core_l3normsq(_d(a[i]), _d(b[i])) == fx['l3sq'][i] for i in range(n))
    assert all(harness.core_distance(_d(a[i]), _d(b[i])) == fx['dist'][i] for i in range(n))
    v = np.ascontiguousarray(fx['v'])
    assert all(harness.core_get_phi(_d(v[i])) == fx['phi'][i] for i in range(n))
    assert all(harness.core_pi_2_pi(float(x)) == y for x, y in zip(fx['ang'], fx['p2p']))
    assert all(harness.core_trunc5(float(x)) == y for x, y in zip(fx['tr_in'], fx['tr']))
    thr = cos_threshold(math.pi / 4)
    vf = np.ascontiguousarray(fx['vf'], np.float32)
    for i in range(n):
        got = harness.core_posture_ok(thr, 0.1, vf[i].ctypes.data_as(C.POINTER(C.c_float)), float(fx['posz'][i]), _d(v[i]))
        assert got == fx['sat'][i]
    pA = np.ascontiguousarray(fx['pA']); pB = np.ascontiguousarray(fx['pB']); vd = np.ascontiguousarray(fx['vd'])
    for i in range(n):
        if fx['isx'][i] == 2:
            continue
        assert harness.core_is_intersect(_d(pA[i]), _d(pB[i]), float(fx['R'][i]), _d(vd[i])) == fx['isx'][i]
    out = np.zeros(7)
    head = np.ascontiguousarray(fx['head']); vv = np.ascontiguousarray(fx['vv'])
    for i in range(n):
        for off, key in ((0, 'c2s'), (1, 'c2s_off')):
            harness.core_c2s(_d(head[i]), _d(vv[i]), off, _d(out))
            # float32 is what the env stores (mampenv.py:31)
            assert np.array_equal(out.astype(np.float32), fx[key][i].astype(np.float32))


def test_candidate_table_matches_reference(tables):
    fx = load('F0_candidates')
    for n in (256, 128):
        unit, _ = tables[n]
        u = unit.reshape(3, n).T
        assert np.array_equal(np.concatenate([0.5 * u, 1.0 * u]), fx[f'cand{n}'])


def test_host_kd_build_matches_oracle(oracle):
    from sca_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for n in (1, 5, 10, 11, 37, 100, 1000):
        pos = rng.uniform(-40, 40, (n, 3))
        if n >= 37:
            pos[: n // 3] = np.round(pos[: n // 3], 0)        # duplicates / ties on the split plane
        perm0 = rng.permutation(n).astype(np.int32)
        p1, p2 = perm0.copy(), perm0.copy()
        t1 = np.zeros((2 * n - 1, 10)); t2 = np.zeros((2 * n - 1, 10))
        oracle.lib().orc_kd_build(n, oracle._d(pos), oracle._p(p1, C.c_int32), oracle._d(t1))
        assert L.sca_kd_build_host(n, _lib.ptr(pos, C.c_double), _lib.ptr(p2, C.c_int32), _lib.ptr(t2, C.c_double)) == 0
        assert np.array_equal(p1, p2)
        used = np.zeros(2 * n - 1, bool)          # compare only reachable nodes

        def walk(i):
            used[i] = True
            if t1[i, 1] - t1[i, 0] > 10:
                walk(int(t1[i, 2])); walk(int(t1[i, 3]))
        walk(0)
        assert np.array_equal(t1[used], t2[used])


@pytest.mark.parametrize('name', episode_fixtures())
def test_core_solve_matches_reference(harness, tables, name):
    fx = load(name)
    st = static_inputs(fx)
    fp = dict(neighbor_dist=10.0, time_step=0.1, time_horizon=10.0, max_speed=1.0, max_heading_change=math.pi / 4)
    fp.update({k: v for k, v in fixture_params(fx)[0].items() if k in fp})          # F16: the parameters the scene was recorded under
    per_agent = fixture_agent_params(fx)                                             # F17: attributes that differ from agent to agent

    def par_of(i):
        q = dict(fp)
        q.update({k: float(v[i]) for k, v in per_agent.items() if k in q})
        return np.array([q['neighbor_dist'], q['time_step'], q['time_horizon'], q['max_speed'], q['max_heading_change'], 0.5,
                         cos_threshold(q['max_heading_change'])])
    T = len(fx['step'])
    n = len(st['radius'])
    u256, p256 = tables[256]
    u128, p128 = tables[128]
    action = np.zeros(7, np.float32)
    vp_out = np.zeros(3)
    diag = np.zeros(5, np.int32)
    fpt, bpt, ipt = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    steps = range(T) if T <= 60 else list(range(0, T, max(1, T // 60)))
    for t in steps:
        pos, vel, head, flags = fx['pos'][t], fx['vel'][t], fx['heading'][t], fx['flags'][t]
        for i in np.nonzero(fx['called'][t])[0]:
            K = int(fx['nbr_n'][t][i]) if fx['nbr_valid'][t][i] else 0
            ids = fx['nbr_id'][t][i][:K]
            kinds = fx['nbr_kind'][t][i][:K].astype(np.uint8)
            nb_pos = np.zeros((max(K, 1), 3)); nb_vel = np.zeros((max(K, 1), 3), np.float32)
            nb_rad = np.zeros(max(K, 1)); nb_goal = np.zeros(max(K, 1), np.uint8); nb_ob = np.zeros(max(K, 1), np.uint8)
            for k in range(K):
                if kinds[k]:
                    nb_pos[k] = st['obs_pos'][ids[k]]; nb_rad[k] = st['obs_radius'][ids[k]]; nb_ob[k] = 1
                else:
                    nb_pos[k] = pos[ids[k]]; nb_vel[k] = vel[ids[k]]; nb_rad[k] = st['radius'][ids[k]]
                    nb_goal[k] = flags[ids[k]] & 1
            given = int(st['vpref_mode'][i])
            vin = np.ascontiguousarray(np.nan_to_num(fx['vpref'][t][i]))
            p_i = np.ascontiguousarray(pos[i]); v_i = np.ascontiguousarray(vel[i], np.float32)
            h_i = np.ascontiguousarray(head[i]); g_i = np.ascontiguousarray(fx['goal'][t][i])
            par = par_of(i)
            stt = harness.core_solve_agent(_d(par), int(st['policy'][i]), int(st['zaxis'][i]), float(st['pref_speed'][i]),
                                           _d(p_i), v_i.ctypes.data_as(fpt), float(st['radius'][i]), _d(h_i), _d(g_i),
                                           given, _d(vin), K, _d(nb_pos), nb_vel.ctypes.data_as(fpt), _d(nb_rad),
                                           nb_ob.ctypes.data_as(bpt), nb_goal.ctypes.data_as(bpt), _d(u256), _d(u128),
                                           _d(p256), _d(p128), action.ctypes.data_as(fpt), _d(vp_out),
                                           diag.ctypes.data_as(ipt))
            assert stt == 0
            if not given:
                assert np.array_equal(vp_out, fx['vpref'][t][i]), (name, t, i, 'v_pref')
            if fx['n_suit'][t][i] >= 0:
                assert diag[0] == fx['n_suit'][t][i] and diag[1] == fx['fallback'][t][i], (name, t, i, diag)
            if fx['plane_fail'][t][i] >= 0:
                assert diag[3] == fx['plane_fail'][t][i] and diag[4] == fx['lp4'][t][i], (name, t, i, diag)
            # velocity + speed bit-exact; the two heading deltas go through atan2 and a cancelling subtraction
            # (x*x instead of pow(x,2) under the sqrt): allow one float32 ulp at |angle| <= 2*pi
            ref = fx['action'][t][i]
            assert np.array_equal(action[:4], ref[:4]), (name, t, i, action, ref)
            assert np.array_equal(action[4:], ref[4:]), (name, t, i, action, ref)      # (atan2 / pow on the restated glibc: equal, round 6)
