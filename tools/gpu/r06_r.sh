#!/bin/bash
# round 6, second session: the split solve (k_solve_sweep beside the re-plans + k_solve_pick4 behind them) forced on / off for the small tracked
# configs, in the driver's form (20 steps after 5) and at 300 steps
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_split
mkdir -p $O
cd $R
for rep in 1 2 3; do
for cfg in "auto SCA_QUIET=1" "split1 SCA_SOLVE_SPLIT=1" "split0 SCA_SOLVE_SPLIT=0"; do
  set -- $cfg
  for w in c5 c2 heldout; do
    env $2 SCA_BENCH_DETAIL=$O/${w}drv_$1_$rep.json timeout 300 python3 bench.py --workload $w --steps 20 --warmup 5 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
    env $2 SCA_BENCH_DETAIL=$O/${w}std_$1_$rep.json timeout 300 python3 bench.py --workload $w --steps 300 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  done
done
done
python3 - $O <<'PY'
import json, os, sys, collections
O = sys.argv[1]
t = collections.defaultdict(dict)
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        leg, v, rep = f[:-5].rsplit('_', 2)
        d = json.load(open(os.path.join(O, f)))
        t[leg].setdefault(v, []).append(d['ms_per_step'])
for leg in t:
    print(leg, '  '.join('%s %s' % (v, '/'.join('%.4f' % x for x in xs)) for v, xs in t[leg].items()))
PY
