#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from the per-launch PMC summaries tools/pmc_summary.py wrote
(profiles/<set>_<workload>_pmc_{sq_counters,fetch_size,write_size}.csv): per kernel the FETCH_SIZE / WRITE_SIZE bytes per
launch, the HBM traffic interval after the calibration (FETCH x 1.17 + WRITE ... FETCH x 2 + WRITE: MI355X_MICROARCH's
rocprofv3 section + tools/fetch_calib.hip), and the VALU wave-instructions per launch; per workload the per-plan / per-agent
instruction figures bench.py prices the issue rate with.  The calibration block is carried over from the previous file.

    python tools/pmc_traffic.py <set prefix, e.g. profiles/r03_b> <out.json> [previous.json] --plans-c4 <re-plans per launch>
"""
import csv
import json
import os
import sys


def table(path):
    if not os.path.exists(path):
        return {}
    return {r['kernel'].replace('sca::', '').split('<')[0]: r for r in csv.DictReader(open(path))}


def workload(prefix, name):
    sq, fe, wr = (table(f'{prefix}_{name}_pmc_{k}.csv') for k in ('sq_counters', 'fetch_size', 'write_size'))
    out = {}
    for k in sorted(set(sq) | set(fe) | set(wr)):
        if k.startswith('__amd') or k.startswith('at::') or ' ' in k:
            continue
        f = float(fe[k]['FETCH_SIZE_per_launch']) * 1024 if k in fe else None            # (KiB in the summaries)
        w = float(wr[k]['WRITE_SIZE_per_launch']) * 1024 if k in wr else None
        if f is not None and w is not None:
            out[f'{k}_fetch_size_bytes'] = round(f)
            out[f'{k}_write_size_bytes'] = round(w)
            out[f'{k}_hbm_bytes_per_launch'] = round(2 * f + w)
            out[f'{k}_hbm_bytes_per_launch_lower'] = round(1.17 * f + w)
        if k in sq:
            out[f'{k}_valu_wave_insts_per_launch'] = float(sq[k]['SQ_INSTS_VALU_per_launch'])
            out[f'{k}_salu_wave_insts_per_launch'] = float(sq[k]['SQ_INSTS_SALU_per_launch'])
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    prefix, dst = args[0], args[1]
    prev = json.load(open(args[2])) if len(args) > 2 else {}
    plans = float(sys.argv[sys.argv.index('--plans-c4') + 1]) if '--plans-c4' in sys.argv else 96300.0
    d = {'calibration': prev.get('calibration', {}),
         'source': f'{prefix}_*_pmc_*.csv: rocprofv3 --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE each in a run of its own) over '
                   '`bench.py --workload <w> --steps 8 --warmup 12 --no-cpu-baseline --no-extra`, per-kernel per-launch means by '
                   'tools/pmc_summary.py (FETCH / WRITE in KiB there, bytes here); this file by tools/pmc_traffic.py'}
    c4 = workload(prefix, 'c4_e2e')
    if 'k_replan_valu_wave_insts_per_launch' in c4:
        c4['k_replan_valu_wave_insts_per_plan'] = c4['k_replan_valu_wave_insts_per_launch'] / plans
        c4['k_replan_plans_per_launch_assumed'] = plans
    c4['k_solve_valu_wave_insts_per_agent'] = prev.get('c4', {}).get('k_solve_valu_wave_insts_per_agent', 816)
    d['c4'] = c4
    c3 = workload(prefix, 'c3')
    if 'k_solve_valu_wave_insts_per_launch' in c3:
        c3['k_solve_valu_wave_insts_per_agent'] = c3['k_solve_valu_wave_insts_per_launch'] / 4096.0
    d['c3'] = c3
    json.dump(d, open(dst, 'w'), indent=1)
    print('wrote', dst, {k: round(v) for k, v in c4.items() if k.startswith('k_replan')})


if __name__ == '__main__':
    main()
