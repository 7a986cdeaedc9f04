#!/bin/bash
# round 6: the whole -m gpu suite on the spin-free launch-free kd query + the two-form K1 kernels, then quick legs
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_g
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q --timeout 400 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "FAILED|passed|failed|Timeout" $O/pytest.log | tail -20
SCA_BENCH_DETAIL=$O/driver_detail.json timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.out 2> $O/driver.err
echo "bench rc $? bytes $(wc -c < $O/driver.out)"; cat $O/driver.out
for tag in tail notail; do
  if [ $tag = notail ]; then export SCA_AUTO_NO_TAIL=1; else unset SCA_AUTO_NO_TAIL; fi
  for w in c3 c3lp; do
    SCA_BENCH_DETAIL=$O/${w}_auto_$tag.json timeout 300 python3 bench.py --workload $w --nbr auto --steps 200 --warmup 30 --no-extra --no-cpu-baseline > $O/${w}_auto_$tag.out 2> $O/${w}_auto_$tag.err
  done
done
unset SCA_AUTO_NO_TAIL
timeout 300 python3 tools/gpu/c4_listed.py > $O/c4_listed.json 2> $O/c4_listed.err
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json') and f.startswith('c3'):
        d = json.load(open(os.path.join(O, f)))
        print('%-28s ms/step %.4f  forms %s' % (f, d['ms_per_step'], d['config'].get('kernel_forms')))
print(open(os.path.join(O, 'c4_listed.json')).read()[:1500])
PY
