"""The native v_pref tracker / 3-D Dubins planner (sca_amd/csrc/sca_dubins.hpp, host code in libsca_hip.so) against
golden vectors recorded from the reference: planner KATs (F7, incl. the paper instance RLRRSL / 976.79 of
dubinsmaneuver3d.py:230) and the per-step v_pref of whole SCA episodes (F1, F2, F4).  Bit-exact."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, fixture_agent_params, fixture_params, fixture_tracker_agent_params, load, static_inputs, tracked_param_fixtures

TRACKED_PARAM_EPISODES = tracked_param_fixtures()


def test_dubins_planner_kats():
    from sca_amd import tracker
    kats = json.load(open(os.path.join(GOLDEN, 'F7_dubins_kat.json')))
    assert kats[0]['mode'] == 'RLRRSL' and abs(kats[0]['length'] - 976.7927) < 1e-3      # the paper instance
    for k in kats:
        length, mode, samples, n = tracker.dubins_plan(k['qi'], k['qf'], k['R'], k['pl'], max_samples=5000)
        assert mode == k['mode'], (mode, k['mode'])
        assert length == k['length']
        assert n == k['n']
        assert np.array_equal(samples[0], k['first'])
        assert np.array_equal(samples[n // 2], k['mid'])
        assert np.array_equal(samples[n - 1], k['last'])


@pytest.mark.parametrize('name', ['F1_sca_circle8', 'F2_sca_circle100', 'F2_rvodubins_circle100', 'F4_sca_takeoff16',
                                  'F4_mixed_takeoff16', 'F10_sca_exp3_map', 'F13_fuzz_track_00', 'F13_fuzz_track_01',
                                  'F13_fuzz_track_02', 'F13_fuzz_track_03', 'F15_sca_circle1024'] + TRACKED_PARAM_EPISODES)
def test_tracker_reproduces_reference_v_pref(name):
    """Open loop on the solver (states come from the fixture), closed loop on the tracker's own state: every
    compute_v_pref of the episode, including all re-plans, must return the reference's V_des bit for bit."""
    from sca_amd import tracker
    fx = load(name)
    st = static_inputs(fx)
    n = len(st['radius'])
    ext = st['vpref_mode'].astype(bool)
    params, trk = fixture_params(fx)                           # F16: turning_radius / pitchlims / neighborDist off their defaults
    tr = tracker.DubinsTracker(fx['goal'][0], fx['goal6'][:, 3:6], st['pref_speed'], st['zaxis'], nthreads=4,
                               neighbor_dist=params.get('neighbor_dist', 10.0), **trk)
    if 'neighbor_dist' in fixture_agent_params(fx):                 # F17: every agent its own neighborDist (scaPolicy.py:299)
        tr.set_neighbor_dist(fixture_agent_params(fx)['neighbor_dist'])
    if fixture_tracker_agent_params(fx):                            # F18: every agent its own turning radius and pitch limits
        tr.set_agent_params(**fixture_tracker_agent_params(fx))
    T = len(fx['step'])
    assert np.array_equal(fx['step'], np.arange(T))           # every step recorded: the tracker state can be replayed
    ever = np.zeros(n, bool)
    for t in range(T):
        active = fx['called'][t].astype(bool) & ext
        ever |= active
        got = tr.vpref(fx['pos'][t], fx['vel'][t], fx['heading'][t], active.astype(np.uint8))
        assert np.array_equal(got[active], fx['vpref'][t][active]), (name, t, np.abs(got[active] - fx['vpref'][t][active]).max())
        tr.note_neighbors(fx['nbr_valid'][t], fx['nbr_n'][t], fx['nbr_dsq'][t])
    assert tr.replans()[ever].min() >= 1                          # (the F16 fuzz scenes hold agents that are done from the start)
    tr.close()


def test_uniform_csc_word_equals_literal_words():
    """The device's four-lanes-per-plan planner evaluates LSL / RSR / LSR / RSL as one sign-parametrised instruction stream
    (sca_dubins.hpp csc_word_uniform); on the host, with glibc on both sides, it must equal the literal words of
    dubinsmaneuver2d.py:33-109 bit for bit -- random frames plus the symmetric ones (alpha = beta, zeros, d = 0)."""
    import ctypes as C
    from sca_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    n = 300000
    al = rng.uniform(0, 2 * np.pi, n)
    be = rng.uniform(0, 2 * np.pi, n)
    d = 10 ** rng.uniform(-3, 4, n)
    al[:1000] = 0
    be[1000:2000] = 0
    al[2000:3000] = be[2000:3000]
    d[3000:4000] = 0
    al[4000:5000] = np.pi
    be[5000:6000] = np.pi / 2
    al[6000:7000] = np.round(al[6000:7000], 2)
    be[6000:7000] = 0.0
    d[7000:8000] = rng.uniform(0, 4, 1000)                     # close end points: every word feasible or nearly so
    bad = C.c_int64(-1)
    assert L.sca_selftest_dubins_words(n, _lib.ptr(al, C.c_double), _lib.ptr(be, C.c_double), _lib.ptr(d, C.c_double),
                                       C.byref(bad)) == 0
    assert bad.value == 0
