"""CPU test of the lean search the device's lane-per-plan kernels run (sca_dubins.hpp, plan3d_lean): compiled for the host, it
must give the literal planner's plan (dubinsmaneuver3d.py:34-113 as restated in plan3d, itself pinned against the reference's
golden vectors by tests/test_tracker.py) bit for bit -- maneuvers, words, candidate count -- on far, level, lopsided, near and
degenerate poses, and most candidates of the far families must really take the lean block."""
import ctypes as C
import math

import numpy as np
import pytest

from sca_amd import _lib


def _run(q, rmin=1.5):
    L = _lib.lib()
    q = np.ascontiguousarray(q, dtype=np.float64)
    bad, lean, lit = C.c_int64(), C.c_int64(), C.c_int64()
    rc = L.sca_selftest_plan3d_lean(len(q), _lib.ptr(q, C.c_double), rmin, -math.pi / 4, math.pi / 4, C.byref(bad), C.byref(lean), C.byref(lit))
    assert rc == 0
    return bad.value, lean.value, lit.value


def _circle(rng, n, R, sigma_xy=1.0, sigma_z=0.5, sigma_yaw=0.3, sigma_pitch=0.1):
    th = rng.uniform(0, 2 * np.pi, n)
    q = np.zeros((n, 10))
    q[:, 0] = R * np.cos(th) + rng.normal(0, sigma_xy, n); q[:, 1] = R * np.sin(th) + rng.normal(0, sigma_xy, n); q[:, 2] = 10 + rng.normal(0, sigma_z, n)
    q[:, 3] = np.mod(th + np.pi + rng.normal(0, sigma_yaw, n), 2 * np.pi); q[:, 4] = rng.normal(0, sigma_pitch, n)
    q[:, 5] = -R * np.cos(th); q[:, 6] = -R * np.sin(th); q[:, 7] = 10; q[:, 8] = np.mod(th + np.pi, 2 * np.pi); q[:, 9] = 0
    return q


FAMILIES = ['far', 'level', 'tiny_dz', 'tiny_pitch', 'medium', 'near', 'mid', 'climb', 'start_pose', 'big_radius']


@pytest.mark.parametrize('family', FAMILIES)
def test_lean_search_equals_the_literal_planner(family):
    rng = np.random.default_rng(FAMILIES.index(family) + 11)
    n = 1500
    rmin = 1.5
    if family == 'far':                                   # the benchmark circle's geometry: paths of 10^4 turning radii
        q = _circle(rng, n, 20000.0)
    elif family == 'level':                               # level flight at the goal's altitude: dz = 0, pitch = 0 exactly
        q = _circle(rng, n, 20000.0); q[:, 2] = 10; q[:, 4] = 0
    elif family == 'tiny_dz':                             # quotients dz / length below 2^-57: glibc's early return in atan2
        q = _circle(rng, n, 20000.0); q[:, 2] = 10 + rng.choice([1e-15, -2e-14, 3e-13, 0.0, 5e-12], n)
    elif family == 'tiny_pitch':                          # cos(beta) - cos(alpha) of the vertical frame tiny or zero
        q = _circle(rng, n, 20000.0); q[:, 4] = rng.choice([1e-9, -1e-8, 1e-7, 0.0, 1e-10, 3e-6], n)
    elif family == 'medium':                              # d between 7 and 36: the table piece of the arctangents
        q = _circle(rng, n, rng.uniform(8.0, 40.0, n) * 1.5)
    elif family == 'near':                                # closer than four radii: CCC words, infeasible words, the literal way
        q = np.zeros((n, 10)); q[:, 0:3] = rng.uniform(-6, 6, (n, 3)); q[:, 5:8] = rng.uniform(-6, 6, (n, 3))
        q[:, 3] = rng.uniform(0, 2 * np.pi, n); q[:, 8] = rng.uniform(0, 2 * np.pi, n); q[:, 4] = rng.uniform(-0.5, 0.5, n); q[:, 9] = rng.uniform(-0.5, 0.5, n)
    elif family == 'mid':
        q = np.zeros((n, 10)); q[:, 0:3] = rng.uniform(-300, 300, (n, 3)); q[:, 5:8] = rng.uniform(-300, 300, (n, 3))
        q[:, 3] = rng.uniform(0, 2 * np.pi, n); q[:, 8] = rng.uniform(0, 2 * np.pi, n); q[:, 4] = rng.uniform(-0.5, 0.5, n); q[:, 9] = rng.uniform(-0.5, 0.5, n)
    elif family == 'climb':                               # altitude differences that need the doubling stage
        q = _circle(rng, n, 300.0); q[:, 2] = 10 + rng.uniform(-400, 400, n)
    elif family == 'start_pose':                          # an episode's first plan: heading = goal heading, level, on the chord
        q = _circle(rng, n, 20000.0, 0.0, 0.0, 0.0, 0.0)
    else:                                                 # another turning radius
        q = _circle(rng, n, 5000.0); rmin = 40.0
    bad, lean, lit = _run(q, rmin)
    assert bad == 0, (family, bad)
    if family in ('far', 'level', 'tiny_dz', 'tiny_pitch', 'start_pose', 'big_radius'):
        assert lean > 5 * lit, (family, lean, lit)        # (the literal way is left for the winner's construction and the odd candidate)
    if family == 'medium':
        assert lean > lit, (family, lean, lit)
