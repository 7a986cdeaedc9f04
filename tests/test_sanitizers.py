"""Sanitizer builds of the CPU side: the oracle under ASan + UBSan, the kernel's arithmetic header
(host harness) under UBSan, the host tracker's thread pool under TSan.  Each runs in a subprocess (the sanitizer runtimes
want to be loaded first / abort on a finding); a finding fails the test with the sanitizer's report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'tests', '_build')


def _runtime(name):
    p = subprocess.run(['gcc', '-print-file-name=' + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _report(r):
    return (r.stdout[-3000:] + '\n' + r.stderr[-6000:])


def test_oracle_under_asan_and_ubsan():
    """golden episodes (single step KATs, the dense-cluster fixtures with collisions / fallback / LP4, fuzz scenes) replayed
    through liboracle_asan.so: no out-of-bounds access, no undefined behaviour, and still bit-exact"""
    asan = _runtime('libasan.so')
    if asan is None:
        pytest.skip('libasan.so not found')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=0',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1', SCA_ORACLE_SO=os.path.join(ROOT, 'oracle', 'liboracle_asan.so'))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_oracle_golden.py'), '-x', '-q', '-p',
                        'no:cacheprovider', '-k', 'candidate or kat or F5 or F12_fuzz_0 or F14 or F4_mixed or F6'],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = _report(r)
    assert r.returncode == 0 and 'AddressSanitizer' not in out and 'runtime error' not in out, out
    assert ' passed' in r.stdout, out


def test_kernel_arithmetic_header_under_ubsan():
    """sca_core.h (what the HIP kernels compute with) compiled for the host with -fsanitize=undefined, -fno-sanitize-recover:
    the harness's golden-vector tests must pass without a report (signed overflow, bad shifts, invalid casts ...)"""
    if _runtime('libubsan.so') is None:
        pytest.skip('libubsan.so not found')
    env = dict(os.environ, SCA_HARNESS_UBSAN='1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_core_harness.py'), '-x', '-q', '-p',
                        'no:cacheprovider'], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = _report(r)
    assert r.returncode == 0 and 'runtime error' not in out, out


def test_host_tracker_thread_pool_under_tsan():
    """the persistent worker pool of the host v_pref tracker (sca_dubins.hpp: Pool, step_all): several steps with 4 and then
    3 threads, compared with the single-threaded run; ThreadSanitizer must stay silent"""
    if _runtime('libtsan.so') is None:
        pytest.skip('libtsan.so not found')
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, 'tsan_tracker')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=thread', '-ffp-contract=off', '-pthread',
                           '-I' + os.path.join(ROOT, 'sca_amd', 'csrc'), '-o', exe, os.path.join(ROOT, 'tests', 'tsan_tracker.cpp')])
    r = subprocess.run([exe], env=dict(os.environ, TSAN_OPTIONS='halt_on_error=1'), capture_output=True, text=True, timeout=900)
    out = _report(r)
    if 'unexpected memory mapping' in out or 'FATAL: ThreadSanitizer' in out:
        pytest.skip('ThreadSanitizer cannot run in this container: ' + out.strip().splitlines()[0])
    assert r.returncode == 0 and 'WARNING: ThreadSanitizer' not in out, out
    assert 'mismatching components' in r.stdout and ' 0 mismatching' in r.stdout, out
