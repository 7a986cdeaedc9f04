#!/usr/bin/env python3
"""gpurun_out/r06_p (written on the GPU box by tools/gpu/r06_profiles.sh) -> profiles/r06_* (tracked).  Run in the build container."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, 'gpurun_out', 'r06_p')
DST = os.path.join(ROOT, 'profiles')


def one(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None          # (gpurun merges into gpurun_out/: an earlier call's files stay beside the new ones)


def main():
    # the compact line (what the driver parses) and the full record beside it
    for a, b in (('bench_c4_default.json', 'r06_a_bench_c4_default.json'), ('bench_c4_driver_cmd.json', 'r06_a_bench_c4_driver_cmd.json'),
                 ('bench_c4_default_detail.json', 'r06_a_bench_c4_default_detail.json'), ('bench_c4_driver_cmd_detail.json', 'r06_a_bench_c4_driver_cmd_detail.json')):
        if os.path.exists(os.path.join(SRC, a)):
            shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
    legs = ['driver_cmd', 'c4_e2e', 'c4_solver', 'c4_grid', 'c2_e2e', 'c5_e2e', 'c3', 'c3_auto', 'c3lp_auto', 'heldout', 'c4_rank8', 'c4_rank8_grid_solver']
    for leg in legs:
        f = one(f'{leg}/**/*_kernel_stats.csv')
        if f:
            shutil.copy(f, os.path.join(DST, f'r06_a_{leg}_kernel_stats.csv'))
    # rocprofv3 timelines (faithful where the step is long: c4; distorted for chains of short kernels: kept for the kernel durations)
    for leg, anchor, name in (('c4_e2e', '^k_track$', 'r06_c4_timeline.json'), ('c4_rank8', '^k_track$', 'r06_c4_rank8_timeline.json'),
                              ('c4_rank8_grid_solver', '^k_grid_count', 'r06_c4_rank8_grid_solver_timeline.json'),
                              ('c3_auto', '^k_grid_count', 'r06_c3_auto_timeline_rocprof.json'), ('c2_e2e', '^k_track_group', 'r06_c2_timeline_rocprof.json')):
        f = one(f'{leg}/**/*_kernel_trace.csv')
        if not f:
            print('no trace for', leg)
            continue
        bench = json.load(open(os.path.join(SRC, leg + '_detail.json')))
        note = (f'rocprofv3 --kernel-trace of `bench.py` leg {leg}: the traced run reported {bench["ms_per_step"]:.4f} ms per step; the kernel trace serialises dispatches, '
                'so chains of short kernels run slower traced than untraced (see profiles/README.md and the *_device_timeline.json files)')
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'timeline.py'), f, '--anchor', anchor, '--steps', '10', '-o', os.path.join(DST, name),
                               '--note', note], stdout=subprocess.DEVNULL)
    for f in glob.glob(os.path.join(SRC, 'dtl_*.json')):
        shutil.copy(f, os.path.join(DST, 'r06_' + os.path.basename(f)[4:-5] + '_device_timeline.json'))
    # traced against untraced: one rank of eight, grid, solver alone
    try:
        un = json.load(open(os.path.join(SRC, 'r8gs_untraced_detail.json')))
        tr = json.load(open(os.path.join(SRC, 'c4_rank8_grid_solver_detail.json')))
        stats = list(csv.DictReader(open(one('c4_rank8_grid_solver/**/*_kernel_stats.csv'))))
        per_step = {r['Name'].split('(')[0].replace('void ', '').replace('sca::', ''): float(r['TotalDurationNs']) / 110.0 / 1e3 for r in stats if 'sca::' in r['Name']}
        json.dump({'what': 'one rank of eight at c4, SCA_NBR_GRID, solver alone (bench.py --emulate-rank-of 8 --nbr grid --vpref straight --steps 100 --warmup 10): the same '
                           'command untraced and under rocprofv3 --kernel-trace --stats',
                   'untraced_ms_per_step': un['ms_per_step'], 'traced_ms_per_step': tr['ms_per_step'],
                   'traced_kernel_sum_us_per_step': round(sum(per_step.values()), 2), 'traced_kernel_us_per_step': {k: round(v, 2) for k, v in per_step.items()},
                   'reading': 'rocprofv3 serialises dispatches (start of a kernel = end of its predecessor in every trace): the traced step equals the sum of the traced '
                              'kernel durations, and every kernel is charged its dispatch latency; untraced, back-to-back launches overlap that latency with the '
                              'predecessor.  Round 4\'s scale_model.solver_only.grid row (0.0573 ms) was a different problem: bench.py ran its solver_only legs with the '
                              'tracker\'s last v_pref frozen (sca_device_tracker_disable left vpref_mode set) -- fixed in round 6, the in-bench row and this command agree'},
                  open(os.path.join(DST, 'r06_rank8_grid_solver_reconcile.json'), 'w'), indent=1)
    except Exception as e:                                   # noqa: BLE001
        print('reconcile:', e)
    print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(DST, 'r06_*'))))


if __name__ == '__main__':
    main()
