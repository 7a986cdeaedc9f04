#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_k2
mkdir -p $O
cd $R
for rep in 1 2; do
for cfg in "tail0 SCA_AUTO_TAIL_MAX=0" "tail2 SCA_AUTO_TAIL_MAX=2" "tail8 SCA_AUTO_TAIL_MAX=8" "notail SCA_AUTO_NO_TAIL=1"; do
  set -- $cfg
  for w in c3 c3lp; do
    env $2 SCA_BENCH_DETAIL=$O/${w}_$1_$rep.json timeout 300 python3 bench.py --workload $w --nbr auto --steps 3000 --warmup 50 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  done
done
done
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f))); print('%-20s %.4f' % (f, d['ms_per_step']))
PY
