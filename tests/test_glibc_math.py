"""sca_amd/csrc/sca_glibc_math.h -- glibc 2.35's sin / cos / atan2 / acos / pow(x, 2) restated operation for operation, so that
the v_pref tracker computes the reference's (Python's math module's) bits on the host and on the device -- against the running
glibc: every result must be the same 64 bits.

The golden fixtures were recorded with this container's libm (Ubuntu GLIBC 2.35, the FMA builds its ifunc resolvers pick on an
AVX2 machine); on a box whose libm is another build the comparison says so instead of failing (the fixtures, not the box's
libm, are the parity target: tests/test_tracker.py replays them through the same functions)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BUILD = os.path.join(HERE, '_build')


def _glibc_is_the_recorded_one():
    try:
        v = os.confstr('CS_GNU_LIBC_VERSION')
    except (ValueError, OSError):
        return False
    flags = open('/proc/cpuinfo').read()
    return v == 'glibc 2.35' and ' fma ' in flags and ' avx2 ' in flags


def test_host_build_equals_running_glibc_bit_for_bit():
    """1e6 x 40 cases (every range of s_sin.c, the table intervals of e_asin.c, the eight paths of e_atan2.c, pow's log / exp
    tables, their boundaries, specials); the same harness was run once with 4e7 arguments per case (1.7e9 evaluations): 0."""
    if not _glibc_is_the_recorded_one():
        pytest.skip('this box runs another libm build than the one the fixtures were recorded with')
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, 'glibc_math_harness')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-mfma', '-ffp-contract=off', '-fno-builtin', '-I', os.path.join(ROOT, 'sca_amd', 'csrc'),
                           os.path.join(HERE, 'glibc_math_harness.cpp'), '-o', exe, '-lm'])
    r = subprocess.run([exe, '1'], capture_output=True, text=True)
    assert r.returncode == 0 and 'TOTAL mismatches 0' in r.stdout, r.stdout[-3000:]


def _host(fn, a, b=None):
    from sca_amd import _lib
    L = _lib.lib()
    a = np.ascontiguousarray(a, np.float64)
    out = np.zeros_like(a)
    bb = None if b is None else _lib.ptr(np.ascontiguousarray(b, np.float64), C.c_double)
    assert L.sca_selftest_libm_host(fn, len(a), _lib.ptr(a, C.c_double), bb, _lib.ptr(out, C.c_double)) == 0
    return out


def test_library_build_equals_python_math():
    """The copy inside libsca_hip.so (what the host tracker calls) against Python's math module, i.e. literally what the
    reference calls (dubinsmaneuver2d.py:33-145 math.sin / cos / atan2 / acos, float ** 2)."""
    if not _glibc_is_the_recorded_one():
        pytest.skip('this box runs another libm build than the one the fixtures were recorded with')
    rng = np.random.default_rng(0)
    n = 60000
    x = np.concatenate([rng.uniform(-7, 7, n // 2), rng.uniform(-3e4, 3e4, n // 2)])
    same = lambda u, v: np.array_equal(np.asarray(u).view(np.uint64), np.asarray(v, np.float64).view(np.uint64))
    assert same(_host(0, x), [math.sin(v) for v in x])
    assert same(_host(1, x), [math.cos(v) for v in x])
    u = rng.uniform(-1, 1, n)
    assert same(_host(2, u), [math.acos(v) for v in u])
    y = rng.uniform(-50, 50, n)
    assert same(_host(3, y, x), [math.atan2(p, q) for p, q in zip(y, x)])
    assert same(_host(4, x), [float(v) ** 2 for v in x])
    assert same(_host(4, x), [math.pow(v, 2) for v in x])
