#!/usr/bin/env python3
"""tools/timeline.py -- per-stream timeline of consecutive resident steps out of a `rocprofv3 --kernel-trace` CSV, and the
critical path of each step, so that the overlap claims of DESIGN.md ("beside", "behind the join", "at the pace of the kd stream")
can be recomputed by a reader from start / end timestamps instead of being inferred from per-kernel average durations.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --workload c4 --no-extra --no-cpu-baseline ...
    python tools/timeline.py gpurun_out/tl/**/*_kernel_trace.csv --anchor k_track --steps 10 -o profiles/r05_c4_timeline.json

A step = [start of the anchor kernel's launch t, start of its launch t+1).  The anchor is the first kernel the step's MAIN stream runs
(`k_track` in tracked passes, `k_kd_gather` otherwise; AUTO passes: `k_grid_count`).  Kernels are attributed to the step in whose
window they START; a kernel the previous step enqueued ahead for this one (the kd build of SCA_NBR_AUTO / the build-ahead of
sca_run_steps) is attributed by `--ahead REGEX:STREAM`-free logic: see `attribute()`.

Critical path (per step): walk backwards from the kernel that ends last; the predecessor of a kernel is the kernel (any stream) with
the latest end <= its start + 2 us (the one it can have been waiting for -- same-stream order or an event); the walk stops at the
step's first kernel.  `gap_us` is the idle time between predecessor end and successor start (launch gap / event latency).  This is the
longest dependent chain the timestamps allow; it is exact for chains of back-to-back kernels and a lower bound on slack otherwise.

Output JSON: {source, anchor, steps: [{t, wall_us, kernels: [{k, stream, queue, start_us, end_us, dur_us, grid, wg, vgpr}],
critical_path: [{k, stream, start_us, end_us, gap_us}], busy_us_by_stream, chip_idle_us}], summary: {...medians...}}.
"""
import argparse
import csv
import glob
import json
import re
import statistics
import sys


def short(name):
    """sca::k_kd_block<768, 384>(sca::DeviceView, ...) -> k_kd_block<768,384>; runtime helpers keep their names"""
    m = re.match(r'(?:void )?(?:sca::)?([A-Za-z_0-9]+)(<[^(]*>)?\(', name)
    if m:
        return m.group(1) + (m.group(2) or '').replace(' ', '').replace('sca::', '')
    return name.split('(')[0].replace('sca::', '')


def load(paths):
    rows = []
    for p in paths:
        for r in csv.DictReader(open(p)):
            if r.get('Kind', 'KERNEL_DISPATCH') != 'KERNEL_DISPATCH':
                continue
            rows.append(dict(k=short(r['Kernel_Name']), stream=int(r.get('Stream_Id', 0) or 0), queue=int(r.get('Queue_Id', 0) or 0),
                             s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']),
                             grid=int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']),
                             wg=int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']),
                             vgpr=int(r.get('VGPR_Count', 0) or 0), lds=int(r.get('LDS_Block_Size', 0) or 0)))
    rows.sort(key=lambda r: r['s'])
    return rows


def windows(rows, anchor, stream=None):
    a = [r for r in rows if re.search(anchor, r['k']) and (stream is None or r['stream'] == stream)]
    return a


def critical_path(ks, slack_ns=2000):
    """ks: kernels of one step sorted by start.  Backwards from the last-ending one."""
    if not ks:
        return []
    cur = max(ks, key=lambda r: r['e'])
    chain = [cur]
    while True:
        preds = [r for r in ks if r is not cur and r['e'] <= cur['s'] + slack_ns and r['s'] < cur['s']]
        if not preds:
            break
        cur = max(preds, key=lambda r: r['e'])
        chain.append(cur)
    chain.reverse()
    return chain


def union_len(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def build(rows, anchor, steps, skip_last=1, anchor_stream=None):
    anc = windows(rows, anchor, anchor_stream)
    if len(anc) < steps + 1 + skip_last:
        raise SystemExit(f'only {len(anc)} launches of /{anchor}/ in the trace: need {steps + 1 + skip_last}')
    # the LAST `steps` complete windows of the trace = inside the timed region (the warm-up and the legs before it come first)
    last = len(anc) - 1 - skip_last
    first = last - steps
    out = []
    for t in range(first, last):
        w0, w1 = anc[t]['s'], anc[t + 1]['s']
        ks = [r for r in rows if w0 <= r['s'] < w1]
        cp = critical_path(ks)
        by_stream = {}
        for r in ks:
            by_stream.setdefault(str(r['stream']), []).append((r['s'], r['e']))
        step = dict(t=t - first, wall_us=(w1 - w0) / 1e3,
                    kernels=[dict(k=r['k'], stream=r['stream'], queue=r['queue'], start_us=round((r['s'] - w0) / 1e3, 2),
                                  end_us=round((r['e'] - w0) / 1e3, 2), dur_us=round((r['e'] - r['s']) / 1e3, 2), grid=r['grid'], wg=r['wg'],
                                  vgpr=r['vgpr']) for r in ks],
                    critical_path=[dict(k=r['k'], stream=r['stream'], start_us=round((r['s'] - w0) / 1e3, 2), end_us=round((r['e'] - w0) / 1e3, 2),
                                        gap_us=round((r['s'] - (cp[i - 1]['e'] if i else w0)) / 1e3, 2)) for i, r in enumerate(cp)],
                    busy_us_by_stream={s: round(union_len(iv) / 1e3, 2) for s, iv in by_stream.items()},
                    chip_idle_us=round(((w1 - w0) - union_len([(max(r['s'], w0), min(r['e'], w1)) for r in ks])) / 1e3, 2))
        step['critical_path_kernel_us'] = round(sum(c['end_us'] - c['start_us'] for c in step['critical_path']), 2)
        step['critical_path_gap_us'] = round(sum(max(c['gap_us'], 0.0) for c in step['critical_path']), 2)
        out.append(step)
    return out


def summarise(steps):
    """medians over the steps: wall, per kernel (name, stream): start / end offset and duration; the most frequent critical path"""
    med = statistics.median
    per = {}
    for st in steps:
        seen = {}
        for k in st['kernels']:
            i = seen.get((k['k'], k['stream']), 0)
            seen[(k['k'], k['stream'])] = i + 1
            per.setdefault((k['k'], k['stream'], i), []).append(k)
    table = []
    for (name, stream, i), v in per.items():
        table.append(dict(k=name, stream=stream, nth=i, seen_in_steps=len(v), start_us=round(med(x['start_us'] for x in v), 2),
                          end_us=round(med(x['end_us'] for x in v), 2), dur_us=round(med(x['dur_us'] for x in v), 2), grid=v[0]['grid'], wg=v[0]['wg']))
    table.sort(key=lambda r: r['start_us'])
    paths = {}
    for st in steps:
        key = ' > '.join(f"{c['k']}@{c['stream']}" for c in st['critical_path'])
        paths.setdefault(key, []).append(st)
    top = max(paths.items(), key=lambda kv: len(kv[1]))
    cp_med = []
    for i, c in enumerate(top[1][0]['critical_path']):
        cp_med.append(dict(k=c['k'], stream=c['stream'], dur_us=round(med(s['critical_path'][i]['end_us'] - s['critical_path'][i]['start_us'] for s in top[1]), 2),
                           gap_us=round(med(s['critical_path'][i]['gap_us'] for s in top[1]), 2)))
    return dict(steps=len(steps), wall_us_median=round(med(s['wall_us'] for s in steps), 2), wall_us_min=round(min(s['wall_us'] for s in steps), 2),
                wall_us_max=round(max(s['wall_us'] for s in steps), 2), chip_idle_us_median=round(med(s['chip_idle_us'] for s in steps), 2),
                critical_path_most_frequent=dict(seen_in_steps=len(top[1]), chain=cp_med,
                                                 kernel_us=round(sum(c['dur_us'] for c in cp_med), 2), gap_us=round(sum(max(c['gap_us'], 0) for c in cp_med), 2)),
                kernels_median=table)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace', nargs='+', help='*_kernel_trace.csv (globs allowed)')
    ap.add_argument('--anchor', required=True, help='regex of the kernel that opens a step on the main stream')
    ap.add_argument('--anchor-stream', type=int, default=None)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--skip-last', type=int, default=1, help='complete windows to leave out at the end of the trace')
    ap.add_argument('-o', '--out', default=None)
    ap.add_argument('--note', default='')
    a = ap.parse_args()
    paths = [p for g in a.trace for p in (glob.glob(g, recursive=True) or [g])]
    rows = load(paths)
    steps = build(rows, a.anchor, a.steps, a.skip_last, a.anchor_stream)
    doc = dict(source=[p.split('gpurun_out/')[-1] for p in paths], anchor=a.anchor, note=a.note, units='microseconds from the step\'s anchor launch',
               summary=summarise(steps), steps=steps)
    txt = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, 'w').write(txt + '\n')
    s = doc['summary']
    print(f"{len(steps)} steps, wall median {s['wall_us_median']} us (min {s['wall_us_min']}, max {s['wall_us_max']}), chip idle {s['chip_idle_us_median']} us")
    cp = s['critical_path_most_frequent']
    print(f"critical path ({cp['seen_in_steps']}/{len(steps)} steps): kernels {cp['kernel_us']} us + gaps {cp['gap_us']} us")
    for c in cp['chain']:
        print(f"   {c['k']:<40} stream {c['stream']}  {c['dur_us']:8.2f} us  (gap before {c['gap_us']:.2f})")
    print('kernels (median start / end / dur):')
    for k in s['kernels_median']:
        print(f"   {k['k']:<40} s{k['stream']} #{k['nth']}  {k['start_us']:8.2f} {k['end_us']:8.2f} {k['dur_us']:8.2f}  grid {k['grid']} wg {k['wg']}  in {k['seen_in_steps']} steps")


if __name__ == '__main__':
    main()
