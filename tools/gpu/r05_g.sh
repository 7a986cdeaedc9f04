#!/bin/bash
# round 5: events as the stop events of the kernels they follow (hipExtLaunchKernelGGL; SCA_EXT_STOP=0: separate records) -- the GPU suite, the
# fuzzers, and every bench leg with and without
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_g
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 300 python3 tools/fuzz_auto.py 79 100 2>&1 | tail -1
timeout 600 python3 tools/fuzz_track.py 79 60 2>&1 | tail -1
one() { python3 bench.py --workload $1 --nbr $2 --steps $3 --warmup $4 --no-extra --no-cpu-baseline --no-env-api 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])'; }
for rep in 1 2; do
  for cfg in "c3 auto 200 50" "c2 kd 200 50" "c5 kd 100 30" "c4 kd 20 5" "c3 kd 200 50" "c3 grid 200 50"; do
    set -- $cfg
    echo "$1 $2 $3/$4 records rep=$rep: $(SCA_EXT_STOP=0 one $1 $2 $3 $4)"
    echo "$1 $2 $3/$4 stop-events rep=$rep: $(one $1 $2 $3 $4)"
  done
done | tee $O/ext_stop_ab.txt
