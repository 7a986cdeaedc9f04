#!/bin/bash
# A/B: the prologue kernels' arctangent inline on the constant tables (this build) against the call (-DSCA_PREP_LIBM=0)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_o
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -k "policy_pass_vs_golden or free_running" --timeout 600 > $O/pytest.log 2>&1
tail -2 $O/pytest.log
cp sca_amd/lib/libsca_hip.so /tmp/lib_new.so
SCA_BUILD_DEFS=-DSCA_PREP_LIBM=0 python3 -m sca_amd.build > $O/build_old.log 2>&1
cp sca_amd/lib/libsca_hip.so /tmp/lib_old.so
for rep in 1 2 3; do
for v in new old; do
  cp /tmp/lib_$v.so sca_amd/lib/libsca_hip.so
  SCA_BENCH_DETAIL=$O/c5_${v}_$rep.json timeout 300 python3 bench.py --workload c5 --steps 100 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_BENCH_DETAIL=$O/c4s_${v}_$rep.json timeout 300 python3 bench.py --vpref straight --steps 60 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_BENCH_DETAIL=$O/c4sg_${v}_$rep.json timeout 300 python3 bench.py --vpref straight --nbr grid --steps 60 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
done
done
cp /tmp/lib_new.so sca_amd/lib/libsca_hip.so
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f))); print('%-22s %.4f' % (f, d['ms_per_step']))
PY
