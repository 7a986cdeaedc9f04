#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_i
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_auto.py tests/test_gpu_multirank.py -q --timeout 300 > $O/pytest1.log 2>&1
tail -3 $O/pytest1.log
SCA_AUTO_TAIL_MAX=1000000000 SCA_AUTO_BACKOFF_DIV=1 timeout 600 python3 tools/fuzz_auto.py 702 60 > $O/fuzz_forced.txt 2>&1; tail -1 $O/fuzz_forced.txt
timeout 600 python3 tools/fuzz_auto.py 701 100 > $O/fuzz_default.txt 2>&1; tail -1 $O/fuzz_default.txt
for rep in 1 2 3; do
for tag in tail notail; do
  if [ $tag = notail ]; then export SCA_AUTO_NO_TAIL=1; else unset SCA_AUTO_NO_TAIL; fi
  for w in c3 c3lp; do
    SCA_BENCH_DETAIL=$O/${w}_auto_${tag}_$rep.json timeout 300 python3 bench.py --workload $w --nbr auto --steps 300 --warmup 30 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  done
done
done
unset SCA_AUTO_NO_TAIL
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        d = json.load(open(os.path.join(O, f)))
        print('%-28s ms/step %.4f' % (f, d['ms_per_step']))
PY
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
python3 tools/device_timeline.py c3 --nbr auto --steps 40 -o $O/dtl_c3_auto.json > $O/dtl_c3_auto.txt 2>&1
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
head -3 $O/dtl_c3_auto.txt; tail -12 $O/dtl_c3_auto.txt
