#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_j
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_auto.py -q -k "form_switches" -s --timeout 400 > $O/pytest.log 2>&1
tail -8 $O/pytest.log
# soak: long bursts
SCA_BENCH_DETAIL=$O/c3_soak.json timeout 300 python3 bench.py --workload c3 --nbr auto --steps 3000 --warmup 50 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
SCA_BENCH_DETAIL=$O/c3lp_soak.json timeout 300 python3 bench.py --workload c3lp --nbr auto --steps 3000 --warmup 50 --no-extra --no-cpu-baseline > /dev/null 2>> $O/err.txt
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in ('c3_soak.json', 'c3lp_soak.json'):
    d = json.load(open(os.path.join(O, f))); print(f, d['ms_per_step'], d['config']['agent_steps_timed'])
PY
