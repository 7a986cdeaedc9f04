"""Builds sca_amd/lib/libsca_hip.so (hand-written HIP kernels + C-ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so then travels with the
repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIBDIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libsca_hip.so')
SOURCES = ['sca_hip.hip']
DEPS = ['sca_hip.hip', 'sca_kernels.hip.h', 'sca_kdbuild.hip.h', 'sca_tracker.hip.h', 'sca_grid.hip.h', 'sca_partition.hip.h', 'sca_dubins.hpp', 'sca_glibc_math.h',
        'sca_glibc_tables.h', 'sca_core.h', os.path.join('..', '..', 'include', 'sca_hip.h')]
# -ffp-contract=off: decisions must follow the reference's unfused arithmetic; fma() is explicit where numpy fuses.
# -Xarch_host -mfma: the host tracker's libm (sca_glibc_math.h) is a chain of fused multiply-adds; without it every one is a libm call
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-fno-builtin-pow', '-Xarch_host', '-mfma',
         '-DSCA_GM_LDS_TABLES']   # the tracker kernels keep atan2's and sin / cos's lookup tables in LDS (sca_glibc_math.h)


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libsca_hip.so')
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_lib(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get('SCA_BUILD_DEFS', '').split()            # experiments: extra -D flags
    cmd = [hipcc()] + FLAGS + extra + ['-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build_lib(force=True, verbose=True))
