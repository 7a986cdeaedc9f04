#!/usr/bin/env python3
"""tools/device_timeline.py -- the UN-PERTURBED timeline of resident steps, stamped on the device (needs a GPU and a DEBUG build):

    SCA_BUILD_DEFS=-DSCA_TIMELINE python -m sca_amd.build
    python tools/device_timeline.py c3 --nbr auto --steps 40 -o profiles/r05_c3_auto_device_timeline.json
    python -m sca_amd.build                                     # back to the product build

Why: rocprofv3's kernel trace serialises dispatches -- fine for c4 (traced step 678 us = untraced 678), but a chain of 5-us kernels runs
~1.7x slower traced (c3 AUTO: 161 us per traced step against 92 untraced), so tools/timeline.py's c3 picture has the right kernel
durations and the wrong gaps.  Here lane 0 of every wavefront of every pass kernel stamps the 100-MHz wall clock when it starts and when
it ends (atomicMin / atomicMax per kernel and step, sca_kernels.hip.h TlScope); one sca_run_steps burst, nothing on the host in between.
Per step: first start / last end of each kernel relative to the step's anchor kernel, the critical path by tools/timeline.py's rule
(predecessor = the kernel with the latest end <= start + 2 us), medians over the steps.  Resolution 10 ns; the two atomics per wavefront
and the spill of two registers in k_solve are the instrumentation's cost (the burst's wall time is printed beside the product build's)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench as B  # noqa: E402
import timeline as TL  # noqa: E402
from sca_amd import _lib, solver as S  # noqa: E402

NAMES = ['k_kd_gather', 'k_kd_top', 'k_kd_lv_swap .. k_kd_level_tail', 'k_kd_block', 'k_grid_count', 'k_grid_fill', 'k_neighbors_grid', 'k_neighbors_kd / kd4',
         'k_neighbors_kd_auto', 'k_solve / k_solve_fb', 'k_solve_sweep', 'k_solve_pick4', 'k_lp / k_solve_lpw', 'k_fallback', 'k_action', 'k_collide_finish',
         'k_goal_flags_others', 'k_track', 're-plan kernels']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('workload', choices=sorted(B.WORKLOADS))
    ap.add_argument('--nbr', default='kd', choices=sorted(B.NBR))
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--straight', action='store_true', help='SCA workloads: the solver alone (no tracker in the step)')
    ap.add_argument('-o', '--out', default=None)
    a = ap.parse_args()
    L = _lib.lib()
    if not hasattr(L, 'sca_debug_timeline_enable'):
        sys.exit('this build of the library has no timeline stamps: SCA_BUILD_DEFS=-DSCA_TIMELINE python -m sca_amd.build')
    L.sca_debug_timeline_enable.argtypes = [C.c_void_p]
    L.sca_debug_timeline_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    w = B.WORKLOADS[a.workload]
    scene = B.build_scene(w, w['n'])
    sol = B.make_solver(S, scene, 0)
    B.reset_state(sol, scene)
    tracked = w['policy'] in ('sca', 'mixed') and not a.straight
    if tracked:
        sol.device_tracker_enable(scene['sc']['goal'][:, 3:6])
    mode = B.NBR[a.nbr]
    steps = min(a.steps, 60)
    sol.run_steps(a.warmup, mode)
    sol.synchronize()
    assert L.sca_debug_timeline_enable(sol.ctx) == 0
    t0 = time.perf_counter()
    sol.run_steps(steps, mode)
    sol.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    buf = np.zeros(len(NAMES) * 64 * 2 + 1024, np.uint64)
    nk, ring = C.c_int(0), C.c_int(0)
    assert L.sca_debug_timeline_read(sol.ctx, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(nk), C.byref(ring)) == 0
    assert nk.value == len(NAMES), (nk.value, len(NAMES))
    tl = buf[:nk.value * ring.value * 2].reshape(nk.value, ring.value, 2)
    rows = []
    for k in range(nk.value):
        for t in range(steps):
            s, e = int(tl[k, t, 0]), int(tl[k, t, 1])
            if s != 0xFFFFFFFFFFFFFFFF and e > 0:
                rows.append(dict(k=NAMES[k], stream=0, queue=0, s=s * 10, e=e * 10, grid=0, wg=0, vgpr=0, lds=0, step=t))
    rows.sort(key=lambda r: r['s'])
    # a step's window: from its first stamped kernel that belongs to the MAIN chain (not the kd build enqueued ahead) to the next step's
    main_first = {}
    for r in rows:
        if r['k'].startswith('k_kd') and a.nbr == 'auto':
            continue
        main_first.setdefault(r['step'], r['s'])
    out_steps = []
    for t in range(1, steps - 1):
        w0, w1 = main_first[t], main_first[t + 1]
        ks = [r for r in rows if r['step'] == t]
        cp = TL.critical_path(ks)
        st = dict(t=t, wall_us=(w1 - w0) / 1e3,
                  kernels=[dict(k=r['k'], stream=0, queue=0, start_us=round((r['s'] - w0) / 1e3, 2), end_us=round((r['e'] - w0) / 1e3, 2),
                                dur_us=round((r['e'] - r['s']) / 1e3, 2), grid=0, wg=0, vgpr=0) for r in ks],
                  critical_path=[dict(k=r['k'], stream=0, start_us=round((r['s'] - w0) / 1e3, 2), end_us=round((r['e'] - w0) / 1e3, 2),
                                      gap_us=round((r['s'] - (cp[i - 1]['e'] if i else min(x['s'] for x in ks))) / 1e3, 2)) for i, r in enumerate(cp)],
                  busy_us_by_stream={}, chip_idle_us=round(((w1 - w0) - TL.union_len([(max(r['s'], w0), min(r['e'], w1)) for r in ks if r['e'] > w0 and r['s'] < w1])) / 1e3, 2))
        out_steps.append(st)
    doc = dict(source='device stamps (SCA_TIMELINE build): first start / last end of each kernel\'s wavefronts, 100-MHz wall clock',
               workload=w['desc'], neighbor_search=a.nbr, tracker_in_step=tracked, steps=steps, burst_wall_ms_per_step_this_build=round(wall_ms, 4),
               note='kernels of one step: the kd build of an SCA_NBR_AUTO pass is enqueued behind the PREVIOUS step\'s integrate stage, so its '
                    'start offsets are negative; spans of kernels launched several times per step (the kd level passes) run from the first '
                    'launch\'s start to the last one\'s end',
               units='microseconds from the step\'s first main-chain kernel', summary=TL.summarise(out_steps), steps_detail=out_steps)
    if a.out:
        open(a.out, 'w').write(json.dumps(doc, indent=1) + '\n')
    s = doc['summary']
    print(f"{a.workload} --nbr {a.nbr}: burst {wall_ms:.4f} ms per step in this build; step windows median {s['wall_us_median']} us (min {s['wall_us_min']}, max {s['wall_us_max']}), "
          f"chip idle {s['chip_idle_us_median']} us")
    cp = s['critical_path_most_frequent']
    print(f"critical path ({cp['seen_in_steps']}/{len(out_steps)} steps): kernels {cp['kernel_us']} us + gaps {cp['gap_us']} us")
    for c in cp['chain']:
        print(f"   {c['k']:<36} {c['dur_us']:8.2f} us  (gap before {c['gap_us']:.2f})")
    print('kernels (median start / end / dur):')
    for k in s['kernels_median']:
        print(f"   {k['k']:<36} {k['start_us']:8.2f} {k['end_us']:8.2f} {k['dur_us']:8.2f}   in {k['seen_in_steps']} steps")
    sol.close()


if __name__ == '__main__':
    main()
