"""The 3-D Dubins planner at BASELINE geometry against the reference's own planner (tests/golden/F7b_dubins_kat_long.npz, recorded by
tools/gen_dubins_kat_long.py from dubinsmaneuver3d.dubinsmaneuver3d, unpatched): 48 plans of c4 (N = 100 000 circle: 4.7 .. 39.8 km,
start poses and perturbed mid-flight poses, height differences down to 1e-14 m), 20 of c2 (N = 1024 circle, 61 .. 412 m) and 24 of
c5 (take-off / landing, 11 .. 17 m) -- length (all 64 bits), the six-letter word, both radii, t / p / q of both 2-D maneuvers,
sample count, and for c2 / c5 EVERY path sample.  This is the link "host tracker = reference at c4 / c2 / c5 scale" that
tests/test_gpu_value_parity.py leans on; the device kernels are held to the same vectors in
tests/test_gpu_tracker.py::test_device_planner_long_range_kats."""
import ctypes as C
import os

import numpy as np

from golden_util import GOLDEN


def load_kats():
    return dict(np.load(os.path.join(GOLDEN, 'F7b_dubins_kat_long.npz')))


def words_of(o):
    """the two three-letter words of a sca_tracker_debug / sca_device_tracker_debug row"""
    w = b''
    for v in (int(o[20]), int(o[21])):
        w += bytes([(v >> 16) & 255, (v >> 8) & 255, v & 255])
    return w


def test_fixture_covers_the_baseline_geometries():
    k = load_kats()
    fam = k['family']
    assert (fam == b'c4_start').sum() >= 16 and (fam == b'c4_mid').sum() >= 24 and len(fam) >= 60
    c4 = np.char.startswith(fam, b'c4')
    assert k['length'][c4].max() > 39000 and k['length'][fam == b'c4_start'].min() > 39790      # ~26 500 turning radii
    assert 400 < k['length'][fam == b'c2_start'].min() < 420 and k['length'][np.char.startswith(fam, b'c5')].max() < 20
    assert (np.abs(k['qf'][c4, 2] - k['qi'][c4, 2]) < 1e-6).sum() >= 5                         # nearly level 40-km plans


def test_host_planner_equals_reference_at_baseline_geometry():
    from sca_amd import tracker
    k = load_kats()
    K = len(k['length'])
    off = k['samples_off']
    for i in range(K):
        keep = off[i + 1] > off[i]
        length, mode, samples, n = tracker.dubins_plan(k['qi'][i], k['qf'][i], float(k['rmin']), tuple(k['pitchlims']),
                                                       max_samples=int(k['n'][i]) if keep else 0)
        assert mode.encode() == k['mode'][i], (i, mode, k['mode'][i])
        assert length == k['length'][i], (i, length, k['length'][i])            # float equality: all 64 bits
        assert n == k['n'][i], (i, n, k['n'][i])
        if keep:
            assert np.array_equal(samples, k['samples'][off[i]:off[i + 1]]), (i, k['family'][i])
    # first / mid / last sample of the 40-km plans (their 1001 samples are not in the fixture)
    for i in np.flatnonzero(np.char.startswith(k['family'], b'c4'))[::3]:
        _, _, samples, n = tracker.dubins_plan(k['qi'][i], k['qf'][i], float(k['rmin']), tuple(k['pitchlims']), max_samples=int(k['n'][i]))
        assert np.array_equal(samples[0], k['first'][i]) and np.array_equal(samples[n // 2], k['mid'][i]) and np.array_equal(samples[n - 1], k['last'][i])


def test_host_tracker_first_plan_has_the_reference_maneuvers():
    """through the tracker (the path the policy takes): the first compute_v_pref plans; radii, t, p, lengths of both 2-D maneuvers"""
    from sca_amd import _lib, tracker
    k = load_kats()
    K = len(k['length'])
    goal = np.ascontiguousarray(k['qf'][:, :3])
    gh = np.concatenate([k['qf'][:, 3:5], np.zeros((K, 1))], 1)
    tr = tracker.DubinsTracker(goal, gh, 1.0, nthreads=4)
    head = np.concatenate([k['qi'][:, 3:5], np.zeros((K, 1))], 1)
    tr.vpref(k['qi'][:, :3], np.zeros((K, 3), np.float32), head, np.ones(K, np.uint8))
    L = _lib.lib()
    o = np.zeros(24)
    for i in range(K):
        assert L.sca_tracker_debug(tr.h, i, _lib.ptr(o, C.c_double)) == 0
        assert o[8] == k['length'][i] and words_of(o) == k['mode'][i], (i, o[8], k['length'][i])
        assert o[0] == k['radii'][i, 0] and o[4] == k['radii'][i, 1], i
        assert o[1] == k['tpq'][i, 0] and o[2] == k['tpq'][i, 1] and o[5] == k['tpq'][i, 3] and o[6] == k['tpq'][i, 4], i
        assert o[9] == k['sampling'][i] and int(o[13]) == k['n'][i], i
    tr.close()


def test_lean_search_equals_literal_planner_on_the_kats():
    """the device's lean search, compiled for the host, on the same poses (the c4 ones go through its far block)"""
    from sca_amd import _lib
    k = load_kats()
    q = np.ascontiguousarray(np.concatenate([k['qi'], k['qf']], 1))
    bad, lean, lit = C.c_int64(-1), C.c_int64(0), C.c_int64(0)
    assert _lib.lib().sca_selftest_plan3d_lean(len(q), _lib.ptr(q, C.c_double), float(k['rmin']), float(k['pitchlims'][0]),
                                               float(k['pitchlims'][1]), C.byref(bad), C.byref(lean), C.byref(lit)) == 0
    assert bad.value == 0 and lean.value > 1000
