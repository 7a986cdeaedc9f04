#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel in libsca_hip.so, read from the gfx950 code object's metadata notes
(llvm-readelf --notes).  Usage: python tools/kernel_regs.py [path/to/lib.so] [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def code_object(lib, out):
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={lib}',
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={out}'], stderr=subprocess.DEVNULL)


def kernels(lib):
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, 'k.co')
        try:
            code_object(lib, co)
        except subprocess.CalledProcessError:
            # a shared library: the bundle sits in the .hip_fatbin section
            fb = os.path.join(d, 'fatbin')
            subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fb])
            code_object(fb, co)
        notes = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], text=True)
    out = []
    for blk in notes.split('- .agpr_count:')[1:]:
        g = lambda k: (re.search(r'\.%s:\s*(\S+)' % k, blk) or [None, '?'])[1]
        out.append(dict(name=g('name'), vgpr=g('vgpr_count'), agpr=blk.split()[0], sgpr=g('sgpr_count'), lds=g('group_segment_fixed_size'),
                        scratch=g('private_segment_fixed_size'), spill=g('vgpr_spill_count'), wg=g('max_flat_workgroup_size')))
    return out


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, 'sca_amd', 'lib', 'libsca_hip.so')
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ''
    print(f'{"kernel":60s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"lds":>7s} {"scratch":>8s} {"spill":>6s} {"wg":>5s}')
    for k in sorted(kernels(lib), key=lambda k: k['name']):
        if flt in k['name']:
            name = subprocess.check_output(['c++filt', k['name']], text=True).strip()
            name = re.sub(r'\(.*', '', name)
            print(f'{name[:60]:60s} {k["vgpr"]:>5s} {k["agpr"]:>5s} {k["sgpr"]:>5s} {k["lds"]:>7s} {k["scratch"]:>8s} {k["spill"]:>6s} {k["wg"]:>5s}')
