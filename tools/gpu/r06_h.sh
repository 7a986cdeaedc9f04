#!/bin/bash
# c3 / c3lp AUTO with and without the launch-free kd query, three runs each (run-to-run scatter)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_h
mkdir -p $O
cd $R
for rep in 1 2 3; do
for tag in tail notail; do
  if [ $tag = notail ]; then export SCA_AUTO_NO_TAIL=1; else unset SCA_AUTO_NO_TAIL; fi
  for w in c3 c3lp; do
    SCA_BENCH_DETAIL=$O/${w}_auto_${tag}_$rep.json timeout 300 python3 bench.py --workload $w --nbr auto --steps 300 --warmup 30 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  done
done
done
unset SCA_AUTO_NO_TAIL
SCA_ACTION_FB_MAX=200000 SCA_BENCH_DETAIL=$O/c4_afb.json timeout 300 python3 bench.py --steps 60 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2>> $O/err.txt
SCA_BENCH_DETAIL=$O/c4_noafb.json timeout 300 python3 bench.py --steps 60 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2>> $O/err.txt
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f)))
        print('%-28s ms/step %.4f  forms %s' % (f, d['ms_per_step'], d['config'].get('kernel_forms')))
PY
