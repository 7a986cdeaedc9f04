"""The DEVICE build of sca_glibc_math.h (what k_track / k_replan* call) against the host build of the same header, which
tests/test_glibc_math.py pins to glibc bit for bit -- and against Python's math module directly on a sample.  Equality of bits:
the device executes the same IEEE-754 add / mul / fma / div / sqrt sequence."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(sol, fn, a, b=None):
    from sca_amd import _lib
    a = np.ascontiguousarray(a, np.float64)
    dev, host = np.zeros_like(a), np.zeros_like(a)
    bb = None if b is None else np.ascontiguousarray(b, np.float64)
    pb = None if bb is None else _lib.ptr(bb, C.c_double)
    assert sol.L.sca_selftest_libm(sol.ctx, fn, len(a), _lib.ptr(a, C.c_double), pb, _lib.ptr(dev, C.c_double)) == 0
    assert sol.L.sca_selftest_libm_host(fn, len(a), _lib.ptr(a, C.c_double), pb, _lib.ptr(host, C.c_double)) == 0
    return dev, host


def _same(u, v):
    u, v = np.asarray(u, np.float64), np.asarray(v, np.float64)
    return ((u.view(np.uint64) == v.view(np.uint64)) | (np.isnan(u) & np.isnan(v)))


def test_device_libm_equals_host_libm_bit_for_bit():
    from sca_amd import solver as S
    rng = np.random.default_rng(11)
    n = 1 << 21
    sol = S.BatchedSolver(max_agents=4)
    mag = lambda lo, hi, m=n: np.exp2(rng.uniform(lo, hi, m)) * rng.choice([-1.0, 1.0], m)
    # sin / cos: all ranges of s_sin.c, their edges, multiples of pi/2
    edges = np.array([0.126, 0.855469, 2.426265, 2 ** -26, 2 ** -27, 1.5707963267948966, 3.141592653589793, 6.283185307179586])
    xs = np.concatenate([rng.uniform(-0.9, 0.9, n), rng.uniform(-6.3, 6.3, n), rng.uniform(-1e3, 1e3, n), rng.uniform(-1.05e8, 1.05e8, n // 2),
                         mag(-60, 26), rng.choice(edges, n // 4) * (1 + rng.uniform(-1e-12, 1e-12, n // 4)) * rng.choice([-1.0, 1.0], n // 4),
                         rng.integers(0, 4000, n // 4) * 1.5707963267948966 + rng.uniform(-1e-9, 1e-9, n // 4),
                         [0.0, -0.0, 5e-324, 1e-300, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-10, 2.2250738585072014e-308]])
    for fn in (0, 1, 5, 6, 9, 10, 12, 13):                       # (12 / 13: update_velocitie's sin / cos, constant tables)
        d, h = _both(sol, fn, xs)
        bad = ~_same(d, h)
        assert not bad.any(), (fn, int(bad.sum()), xs[bad][:4], d[bad][:4], h[bad][:4])
    # acos
    us = np.concatenate([rng.uniform(-1, 1, 2 * n), (1 - np.exp2(rng.uniform(-52, -3, n))) * rng.choice([-1.0, 1.0], n), mag(-60, 0),
                         rng.choice([0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875], n // 4) * (1 + rng.uniform(-1e-13, 1e-13, n // 4)),
                         [0.0, -0.0, 1.0, -1.0, 1.0000000000000002, -1.5, np.nan, np.inf, 0.9999999999999999, -0.9999999999999999]])
    d, h = _both(sol, 2, us)
    bad = ~_same(d, h)
    assert not bad.any(), (int(bad.sum()), us[bad][:4], d[bad][:4], h[bad][:4])
    # pow(x, 2)
    ps = np.concatenate([rng.uniform(-100, 100, 2 * n), rng.uniform(-2, 2, n), mag(-60, 60), mag(-359, 359), 1 + rng.uniform(-1e-6, 1e-6, n) * np.exp2(rng.uniform(-40, 0, n)),
                         [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 3.0]])
    for fn in (4, 8, 14):
        d, h = _both(sol, fn, ps)
        bad = ~_same(d, h)
        assert not bad.any(), (fn, int(bad.sum()), ps[bad][:4], d[bad][:4], h[bad][:4])
    # atan2: magnitudes over hundreds of decades, every sign combination, ratios at the table's break points, the constants the planner passes
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310, 1e308, -1e308, 2.2250738585072014e-308, 1e-160, 1e160, 2.0, 0.0625, 16.0])
    ys = [rng.uniform(-3, 3, 2 * n), rng.uniform(-1e5, 1e5, n), mag(-40, 40), mag(-1000, 1000), rng.choice([-2.0, 2.0, 0.0, -0.0], n), np.repeat(sp, len(sp))]
    xq = [rng.uniform(-3, 3, 2 * n), rng.uniform(-1e5, 1e5, n), mag(-40, 40), mag(-1000, 1000), rng.uniform(0, 5e4, n), np.tile(sp, len(sp))]
    x5 = mag(-3, 3)
    r5 = (rng.integers(1, 257, n) / 256.0) * (1 + rng.uniform(-1e-13, 1e-13, n))
    sw = rng.random(n) < 0.5
    ys.append(np.where(sw, x5 * r5, x5) * rng.choice([-1.0, 1.0], n))
    xq.append(np.where(sw, x5, x5 * r5))
    x6 = mag(-20, 20)
    ys.append(x6 * np.exp2(rng.uniform(-70, 70, n)) * rng.choice([-1.0, 1.0], n))
    xq.append(x6)
    ya, xa = np.concatenate(ys), np.concatenate(xq)
    for fn in (3, 7, 11):                                        # (11: cartesian2spherical's / get_phi's atan2)
        d, h = _both(sol, fn, ya, xa)
        bad = ~_same(d, h)
        assert not bad.any(), (fn, int(bad.sum()), ya[bad][:4], xa[bad][:4], d[bad][:4], h[bad][:4])
    sol.close()


def test_device_libm_equals_python_math_on_a_sample():
    """Directly against what the reference calls (skipped where the box's libm is not the recorded build: the host copy, which
    the test above compares with, is pinned by the fixtures either way)."""
    import os
    try:
        ok = os.confstr('CS_GNU_LIBC_VERSION') == 'glibc 2.35' and ' fma ' in open('/proc/cpuinfo').read()
    except (ValueError, OSError):
        ok = False
    if not ok:
        pytest.skip('another libm build than the recorded one')
    from sca_amd import solver as S
    rng = np.random.default_rng(5)
    n = 40000
    sol = S.BatchedSolver(max_agents=4)
    x = np.concatenate([rng.uniform(-7, 7, n // 2), rng.uniform(-4e4, 4e4, n // 2)])
    y = rng.uniform(-50, 50, n)
    u = rng.uniform(-1, 1, n)
    assert _same(_both(sol, 0, x)[0], [math.sin(v) for v in x]).all()
    assert _same(_both(sol, 1, x)[0], [math.cos(v) for v in x]).all()
    assert _same(_both(sol, 2, u)[0], [math.acos(v) for v in u]).all()
    assert _same(_both(sol, 3, y, x)[0], [math.atan2(p, q) for p, q in zip(y, x)]).all()
    assert _same(_both(sol, 4, x)[0], [float(v) ** 2 for v in x]).all()
    assert _same(_both(sol, 11, y, x)[0], [math.atan2(p, q) for p, q in zip(y, x)]).all()
    assert _same(_both(sol, 12, x)[0], np.sin(x)).all() and _same(_both(sol, 13, x)[0], np.cos(x)).all()      # mampenv.py:91-93 calls numpy's
    assert _same(_both(sol, 14, x)[0], [np.float64(v) ** 2 for v in x]).all()      # the scalar power: libm's pow (an array ** 2 is x * x)
    sol.close()
