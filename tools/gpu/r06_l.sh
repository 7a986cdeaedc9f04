#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_l
mkdir -p $O
cd $R
for rep in 1 2; do
for cfg in "afb16384 SCA_ACTION_FB_MAX=16384" "afb0 SCA_ACTION_FB_MAX=0"; do
  set -- $cfg
  env $2 SCA_BENCH_DETAIL=$O/c3_$1_$rep.json timeout 300 python3 bench.py --workload c3 --nbr auto --steps 3000 --warmup 50 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  env $2 SCA_BENCH_DETAIL=$O/c3kd_$1_$rep.json timeout 300 python3 bench.py --workload c3 --nbr kd --steps 2000 --warmup 50 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  env $2 SCA_BENCH_DETAIL=$O/c5_$1_$rep.json timeout 300 python3 bench.py --workload c5 --steps 400 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
done
done
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f))); print('%-24s %.4f' % (f, d['ms_per_step']))
PY
