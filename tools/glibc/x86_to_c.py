#!/usr/bin/env python3
"""Study aid: renders the scalar-double instructions of a libm function (objdump -d text) as C-like statements, with
rip-relative constants resolved to their values.  Used to read off the exact operation sequence (which a*b+c were fused into
fma by the compiler that built this libm) when restating glibc's sin / cos / atan2 / acos / pow for the device
(sca_amd/csrc/sca_glibc_math.h).  Not part of the product or of the tests.
  python tools/glibc/x86_to_c.py START STOP [libm]"""
import re
import struct
import subprocess
import sys

LIBM = '/lib/x86_64-linux-gnu/libm.so.6'


def sections(lib):
    out = subprocess.check_output(['readelf', '-S', '-W', lib], text=True)
    secs = []
    for m in re.finditer(r'\]\s+(\S+)\s+\S+\s+([0-9a-f]{16})\s+([0-9a-f]{6,})\s+([0-9a-f]{6,})', out):
        secs.append((m.group(1), int(m.group(2), 16), int(m.group(3), 16), int(m.group(4), 16)))
    return secs


class Mem:
    def __init__(self, lib):
        self.data = open(lib, 'rb').read()
        self.secs = sections(lib)

    def off(self, va):
        for name, addr, off, size in self.secs:
            if addr <= va < addr + size and addr:
                return off + va - addr
        raise KeyError(hex(va))

    def f64(self, va):
        return struct.unpack_from('<d', self.data, self.off(va))[0]

    def u64(self, va):
        return struct.unpack_from('<Q', self.data, self.off(va))[0]


def main():
    start, stop = int(sys.argv[1], 16), int(sys.argv[2], 16)
    lib = sys.argv[3] if len(sys.argv) > 3 else LIBM
    mem = Mem(lib)
    txt = subprocess.check_output(['objdump', '-d', '--no-show-raw-insn', f'--start-address={start}', f'--stop-address={stop}', lib], text=True)
    targets = set()
    lines = []
    for ln in txt.splitlines():
        m = re.match(r'\s*([0-9a-f]+):\s+(\S+)\s*(.*)', ln)
        if not m:
            continue
        addr, op, args = int(m.group(1), 16), m.group(2), m.group(3)
        args = re.sub(r'\s*<.*?>', '', args)
        cm = re.search(r'#\s*([0-9a-f]+)', args)
        const = int(cm.group(1), 16) if cm else None
        args = re.sub(r'\s*#.*', '', args).strip()
        lines.append((addr, op, args, const))
        if op.startswith('j') or op == 'call':
            t = re.match(r'([0-9a-f]+)', args)
            if t:
                targets.add(int(t.group(1), 16))

    def opnd(a, const):
        a = a.strip()
        if '(%rip)' in a and const is not None:
            try:
                v = mem.f64(const)
                return f'K[{const:x}]={v!r}(0x{mem.u64(const):016x})'
            except KeyError:
                return f'MEM[{const:x}]'
        return a.replace('%', '')

    for addr, op, args, const in lines:
        lab = f'L{addr:x}:' if addr in targets else ''
        a = [x for x in re.split(r',(?![^(]*\))', args)] if args else []
        o = [opnd(x, const) for x in a]
        s = None
        if op in ('vmulsd', 'vaddsd', 'vsubsd', 'vdivsd') and len(o) == 3:
            sym = {'vmulsd': '*', 'vaddsd': '+', 'vsubsd': '-', 'vdivsd': '/'}[op]
            s = f'{o[2]} = {o[1]} {sym} {o[0]}'
        elif re.match(r'vf(n?)m(add|sub)(132|213|231)sd', op) and len(o) == 3:
            m = re.match(r'vf(n?)m(add|sub)(132|213|231)sd', op)
            neg, kind, order = m.group(1), m.group(2), m.group(3)
            x3, x2, x1 = o[0], o[1], o[2]                      # AT&T: op3, op2, op1(dest)
            if order == '132':
                a_, b_, c_ = x1, x3, x2
            elif order == '213':
                a_, b_, c_ = x2, x1, x3
            else:
                a_, b_, c_ = x2, x3, x1
            sgn = '-' if neg else ''
            cs = '-' if kind == 'sub' else ''
            s = f'{x1} = fma({sgn}{a_}, {b_}, {cs}{c_})'
        elif op in ('vmovsd', 'vmovq', 'vmovapd', 'movsd', 'movapd', 'movq'):
            s = f'{o[-1]} = {o[0] if len(o) == 2 else o[1]}'
        elif op in ('vandpd', 'vorpd', 'vxorpd', 'vandnpd') and len(o) == 3:
            sym = {'vandpd': '&', 'vorpd': '|', 'vxorpd': '^', 'vandnpd': '&~'}[op]
            s = f'{o[2]} = {o[1]} {sym} {o[0]}'
        if s is None:
            s = f'{op} {", ".join(o)}'
        print(f'{lab:10s} {addr:x}: {s}')


if __name__ == '__main__':
    main()
