#!/usr/bin/env python3
"""Per-kernel, per-launch means of a rocprofv3 --pmc run (its *_counter_collection.csv) as one small CSV: the form the files
profiles/*_pmc_*.csv are kept in.

    python tools/pmc_summary.py <dir or counter_collection.csv> <out.csv>
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main(src, out):
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no *_counter_collection.csv under {src}')
    per = defaultdict(lambda: defaultdict(list))            # kernel -> counter -> [value per dispatch]
    for f in files:
        acc = defaultdict(float)                            # (dispatch, kernel, counter) -> sum over the rows of that dispatch
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').strip()
            acc[(r['Dispatch_Id'], name, r['Counter_Name'])] += float(r['Counter_Value'])
        for (_, name, ctr), v in acc.items():
            per[name][ctr].append(v)
    counters = sorted({c for k in per.values() for c in k})
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'launches'] + [c + '_per_launch' for c in counters])
        for name in sorted(per):
            n = max(len(v) for v in per[name].values())
            w.writerow([name, n] + [round(sum(per[name][c]) / max(len(per[name][c]), 1), 2) if c in per[name] else '' for c in counters])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
