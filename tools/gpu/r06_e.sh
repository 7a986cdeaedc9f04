#!/bin/bash
# round 6, fifth GPU call: the tail form of the AUTO kd query + k_action_fb: their tests, the AUTO / solve_fb / grid suites, c3 legs before / after
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_auto.py tests/test_gpu_solve_fb.py -q -x --timeout 900 > $O/pytest1.log 2>&1
tail -5 $O/pytest1.log
for w in c3 c3lp; do
  for tag in "tail" "notail"; do
    if [ $tag = notail ]; then export SCA_AUTO_NO_TAIL=1; else unset SCA_AUTO_NO_TAIL; fi
    for afb in 16384 0; do
      SCA_ACTION_FB_MAX=$afb SCA_BENCH_DETAIL=$O/${w}_${tag}_afb$afb.json python3 bench.py --workload $w --nbr auto --steps 200 --warmup 30 --no-extra --no-cpu-baseline > $O/${w}_${tag}_afb$afb.out 2> $O/${w}_${tag}_afb$afb.err
    done
  done
done
unset SCA_AUTO_NO_TAIL
for afb in 16384 0; do
  SCA_ACTION_FB_MAX=$afb SCA_BENCH_DETAIL=$O/c3_kd_afb$afb.json python3 bench.py --workload c3 --nbr kd --steps 200 --warmup 30 --no-extra --no-cpu-baseline > $O/c3_kd_afb$afb.out 2> $O/c3_kd_afb$afb.err
  SCA_ACTION_FB_MAX=$afb SCA_BENCH_DETAIL=$O/c5_kd_afb$afb.json python3 bench.py --workload c5 --nbr kd --steps 100 --warmup 30 --no-extra --no-cpu-baseline > $O/c5_kd_afb$afb.out 2> $O/c5_kd_afb$afb.err
done
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f)))
        print('%-28s ms/step %.4f  forms %s' % (f, d['ms_per_step'], d['config'].get('kernel_forms')))
PY
