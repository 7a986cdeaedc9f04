#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_m
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_solve_split.py tests/test_gpu_value_parity.py tests/test_gpu_tracker.py -q --timeout 600 > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for rep in 1 2 3; do
for pa in 1 0; do
  SCA_PICK_ACTION=$pa SCA_BENCH_DETAIL=$O/c4_pa${pa}_$rep.json timeout 300 python3 bench.py --steps 60 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_PICK_ACTION=$pa SCA_BENCH_DETAIL=$O/c4drv_pa${pa}_$rep.json timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
done
done
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f))); print('%-22s %.4f  %s' % (f, d['ms_per_step'], d['config']['kernel_forms']))
PY
