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
import sys

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
    # 11-14: the entry points of the policy epilogue and the env update (sca_core.h m_atan2 / m_sincos / m_pow2): util.py:48-49,150-152
    # call math.atan2, mampenv.py:91-94 numpy's sin / cos and np.float64 ** 2
    assert same(_host(11, y, x), [math.atan2(p, q) for p, q in zip(y, x)])
    assert same(_host(12, x), np.sin(x)) and same(_host(13, x), np.cos(x))
    assert same(_host(14, x), [np.float64(v) ** 2 for v in x])          # the SCALAR power (libm's pow; an array ** 2 is x * x)


def test_libm_check_says_what_the_parity_claim_is_conditional_on(tmp_path):
    """sca_libm_check: equal to the running libm on the recorded build -- and, with a libm whose sine is off by one ulp preloaded
    in front of it (what another glibc build looks like from here), it says so: return 1, the sine's count, one note on stderr from
    sca_tracker_create.  Nothing else changes: the tracker keeps computing the restated build's bits."""
    from sca_amd import tracker
    ok, bad = tracker.libm_check()
    if _glibc_is_the_recorded_one():
        assert ok and bad == [0, 0, 0, 0, 0]
    shim = tmp_path / 'offsin.c'
    shim.write_text('#define _GNU_SOURCE\n#include <dlfcn.h>\n#include <math.h>\n'
                    'double sin(double x) { static double (*real)(double); if (!real) real = (double (*)(double))dlsym(RTLD_NEXT, "sin");\n'
                    '  double r = real(x); return nextafter(r, 10.0); }\n')
    so = tmp_path / 'liboffsin.so'
    subprocess.check_call(['gcc', '-O1', '-shared', '-fPIC', '-fno-builtin', str(shim), '-o', str(so), '-ldl', '-lm'])
    code = ('import sys; sys.path.insert(0, %r)\n'
            'import numpy as np\n'
            'from sca_amd import tracker\n'
            'ok, bad = tracker.libm_check(); print("CHECK", ok, bad)\n'
            'tr = tracker.DubinsTracker(np.zeros((1, 3)), np.zeros((1, 3)), 1.0, nthreads=1); tr.close()\n'
            'l, mode, _, n = tracker.dubins_plan([0.0, 0.0, 3.0, -1.5707963267948966, 0.0], [0.0, 0.0, 13.0, 1.5707963267948966, 0.0], 1.5)\n'
            'print("PLAN", mode, repr(l))\n' % ROOT)
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, LD_PRELOAD=str(so)), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('CHECK')][0]
    assert 'False' in line and '[4096, 0, 0, 0, 0]' in line, line
    assert "this host's libm differs from the restated glibc 2.35" in r.stderr
    assert 'PLAN RLRLSR 16.442132978487617' in r.stdout        # the __main__ instance of dubinsmaneuver3d.py: still the recorded bits
