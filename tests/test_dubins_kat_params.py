"""The 3-D Dubins planner away from the reference's default turning radius and pitch limits (tests/golden/F7c_dubins_kat_params.npz,
recorded by tools/gen_dubins_kat_params.py from dubinsmaneuver3d.dubinsmaneuver3d, unpatched): six (Rmin, pitchlims) sets --
Rmin 0.8 / 3 / 10, pitchlims -+pi/4, -+pi/6, (-0.5, 0.9), (-0.2, 0.2) -- ten plans each, end points 1 .. 60 turning radii apart, level,
climbing inside and beyond the limits, straight above each other, mid-flight poses.  Length (all 64 bits), word, radii, t / p / q,
sampling size, sample count and every path sample.  The device kernels are held to the same vectors in
tests/test_gpu_tracker.py::test_device_planner_param_kats."""
import ctypes as C
import os

import numpy as np
import pytest

from golden_util import GOLDEN
from test_dubins_kat_long import words_of


def load_kats():
    return dict(np.load(os.path.join(GOLDEN, 'F7c_dubins_kat_params.npz')))


def test_fixture_covers_the_parameter_sets():
    k = load_kats()
    assert sorted(set(k['set_rmin'])) == [0.8, 3.0, 10.0] and len(k['set_rmin']) == 6
    assert len(k['length']) == 60 and (np.bincount(k['set']) == 10).all()
    d = np.linalg.norm(k['qf'][:, :3] - k['qi'][:, :3], axis=1) / k['set_rmin'][k['set']]
    assert (d < 7).sum() >= 15 and (d > 7).sum() >= 30            # both sides of the lean search's far block
    assert (k['radii'].max(axis=1) > 2 * k['set_rmin'][k['set']]).sum() >= 20      # the radius search had to leave Rmin


def test_host_planner_equals_reference_off_the_default_parameters():
    from sca_amd import tracker
    k = load_kats()
    off = k['samples_off']
    for i in range(len(k['length'])):
        s = k['set'][i]
        length, mode, samples, n = tracker.dubins_plan(k['qi'][i], k['qf'][i], float(k['set_rmin'][s]), tuple(k['set_pitchlims'][s]),
                                                       max_samples=int(k['n'][i]))
        assert mode.encode() == k['mode'][i], (i, mode, k['mode'][i])
        assert length == k['length'][i], (i, length, k['length'][i])
        assert n == k['n'][i], (i, n, k['n'][i])
        assert np.array_equal(samples, k['samples'][off[i]:off[i + 1]]), i


@pytest.mark.parametrize('s', range(6))
def test_host_tracker_first_plan_off_the_default_parameters(s):
    from sca_amd import _lib, tracker
    k = load_kats()
    sel = np.flatnonzero(k['set'] == s)
    K = len(sel)
    qi, qf = k['qi'][sel], k['qf'][sel]
    goal = np.ascontiguousarray(qf[:, :3])
    gh = np.concatenate([qf[:, 3:5], np.zeros((K, 1))], 1)
    tr = tracker.DubinsTracker(goal, gh, 1.0, nthreads=2, turning_radius=float(k['set_rmin'][s]), pitchlims=tuple(k['set_pitchlims'][s]))
    head = np.concatenate([qi[:, 3:5], np.zeros((K, 1))], 1)
    tr.vpref(qi[:, :3], np.zeros((K, 3), np.float32), head, np.ones(K, np.uint8))
    L = _lib.lib()
    o = np.zeros(24)
    for j, i in enumerate(sel):
        assert L.sca_tracker_debug(tr.h, j, _lib.ptr(o, C.c_double)) == 0
        assert o[8] == k['length'][i] and words_of(o) == k['mode'][i], (i, o[8], k['length'][i])
        assert o[0] == k['radii'][i, 0] and o[4] == k['radii'][i, 1], i
        assert o[1] == k['tpq'][i, 0] and o[2] == k['tpq'][i, 1] and o[5] == k['tpq'][i, 3] and o[6] == k['tpq'][i, 4], i
        assert o[9] == k['sampling'][i] and int(o[13]) == k['n'][i], i
    tr.close()


@pytest.mark.parametrize('s', range(6))
def test_lean_search_equals_literal_planner_off_the_default_parameters(s):
    """the device's lean search, compiled for the host, on the same poses and on 4000 random ones per set (the planner's literal form is
    what the first test pins to the reference)"""
    from sca_amd import _lib
    k = load_kats()
    sel = np.flatnonzero(k['set'] == s)
    R, PL = float(k['set_rmin'][s]), k['set_pitchlims'][s]
    rng = np.random.default_rng(700 + s)
    m = 4000
    d = R * 10 ** rng.uniform(-0.5, 2.2, m)
    az = rng.uniform(0, 2 * np.pi, m)
    p0 = rng.uniform(-30, 30, (m, 3)) + np.array([0, 0, 80.0])
    dz = d * np.tan(rng.uniform(-1.2, 1.2, m)) * (rng.random(m) < 0.7)
    p1 = p0 + np.stack([d * np.cos(az), d * np.sin(az), dz], 1)
    up = rng.random(m) < 0.1
    p1[up, :2] = p0[up, :2]
    qr = np.concatenate([p0, rng.uniform(0, 2 * np.pi, (m, 1)), rng.uniform(PL[0], PL[1], (m, 1)),
                         p1, rng.uniform(0, 2 * np.pi, (m, 1)), rng.uniform(PL[0], PL[1], (m, 1))], 1)
    q = np.ascontiguousarray(np.concatenate([np.concatenate([k['qi'][sel], k['qf'][sel]], 1), qr]))
    bad, lean, lit = C.c_int64(-1), C.c_int64(0), C.c_int64(0)
    assert _lib.lib().sca_selftest_plan3d_lean(len(q), _lib.ptr(q, C.c_double), R, float(PL[0]), float(PL[1]), C.byref(bad), C.byref(lean),
                                               C.byref(lit)) == 0
    assert bad.value == 0, (s, bad.value)
    assert lean.value > 1000
