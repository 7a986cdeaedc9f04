#!/usr/bin/env python3
"""Static instruction mix of one kernel of libsca_hip.so (llvm-objdump -d on the gfx950 code object), by issue class -- the classes
tools/valu_calib.hip measures cycles for.  A STATIC histogram weights every instruction of the kernel's body once; k_replan's body is
dominated by its search loop (one candidate = one trip), so the static mix of the loop is what a wavefront issues per candidate.
Usage: python tools/isa_histogram.py k_replan [lib.so]   ->  JSON on stdout"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from kernel_regs import LLVM, code_object  # noqa: E402

CLASSES = [
    # (class, regex, the calibration row of tools/valu_calib.hip that prices it)
    ('valu_f64_fma', r'^v_(fma|fmac)_f64', 'v_fma_f64'),
    ('valu_f64_mul', r'^v_(mul|ldexp)_f64|^v_pk_(mul|fma)_f64', 'v_mul_f64'),
    ('valu_f64_add', r'^v_(add|sub|min|max|fract|trunc|floor|ceil|rndne|frexp_mant)_f64|^v_(min|max)(imum|imum3|3)?_f64|^v_pk_add_f64', 'v_add_f64'),
    ('valu_f64_trans', r'^v_(rcp|rsq|sqrt|div_fmas|div_fixup|div_scale|trig_preop)_f64', 'v_rcp_f64'),
    ('valu_f64_cmp', r'^v_cmpx?_\w+_f64|^v_cmp_class_f64', 'v_cmp_lt_f64'),
    ('valu_cvt', r'^v_cvt_', 'v_cmp_lt_f64'),
    ('valu_select', r'^v_cndmask_b32', 'v_cndmask_b32(sgpr mask)'),
    ('valu_mov', r'^v_(mov_b32|mov_b64|readfirstlane_b32|readlane_b32|writelane_b32|accvgpr|swap|permlane|bfrev|pk_mov)', 'v_mov_b32'),
    ('valu_int_half_rate', r'^v_(lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|mul_lo|mul_hi|mad_u64|mad_i64|mad_u32|mad_i32|bfe|bfi|bfm|alignbit|alignbyte|perm|cmp|cmpx|mbcnt|ffb|lshl_add|add_lshl|lshl_or|frexp_exp|sad|bitop3)\w*', 'v_lshlrev_b32'),
    ('valu_int_full_rate', r'^v_(add|sub|subrev|addc|subb|subbrev|and|or|xor|not|min|max|med3|add3|and_or|or3|xad|xor3)\w*', 'v_add_u32'),
    ('lds', r'^ds_', None),
    ('vmem', r'^(global|flat|buffer|scratch)_', None),
    ('salu', r'^s_(?!waitcnt|nop|endpgm|barrier|setprio|sleep|sethalt|branch|cbranch|code_end)', None),
    ('branch', r'^s_(branch|cbranch)', None),
    ('wait_misc', r'^s_(waitcnt|nop|barrier|setprio|sleep|endpgm)', None),
]


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, 'k.co')
        fb = os.path.join(d, 'fatbin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fb])
        code_object(fb, co)
        return subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], text=True)


def histogram(text, kernel):
    out, inside, total = {}, False, 0
    other = {}
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:', line)
        if m:
            name = subprocess.check_output(['c++filt', m.group(1)], text=True).strip()
            inside = re.match(r'(void )?(sca::)?' + re.escape(kernel) + r'(<[^(]*>)?\(', name) is not None
            continue
        if not inside:
            continue
        ins = line.strip().split('//')[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        total += 1
        for cls, pat, _ in CLASSES:
            if re.match(pat, op):
                out[cls] = out.get(cls, 0) + 1
                break
        else:
            other[op] = other.get(op, 0) + 1
            out['other'] = out.get('other', 0) + 1
    return total, out, other


if __name__ == '__main__':
    kernel = sys.argv[1] if len(sys.argv) > 1 else 'k_replan'
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'sca_amd', 'lib', 'libsca_hip.so')
    total, hist, other = histogram(disassemble(lib), kernel)
    valu = sum(v for k, v in hist.items() if k.startswith('valu'))
    out = {'kernel': kernel, 'static_instructions': total, 'by_class': hist, 'valu_static': valu,
           'valu_mix': {k: round(v / max(valu, 1), 4) for k, v in hist.items() if k.startswith('valu')},
           'priced_by': {c: row for c, _, row in CLASSES if row}, 'unclassified_top': dict(sorted(other.items(), key=lambda kv: -kv[1])[:12])}
    print(json.dumps(out, indent=1))
