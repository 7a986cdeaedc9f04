#!/bin/bash
# round 6, second session: the library built with other instruction-scheduling strategies of the AMDGPU back end (tools/_build/variants/*.so:
# -mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause / iterative-ilp, -amdgpu-schedule-metric-bias=0) against the shipped build: c4 in
# the driver's form and at 100 steps, c2 / c3 AUTO / c5 at 300 steps.  Scheduling does not change a single result bit (no reassociation).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUTDIR:-r06_sched}
mkdir -p $O
cd $R
cp sca_amd/lib/libsca_hip.so /tmp/base.so
for rep in 1 2; do
for v in ${VARIANTS:-base ilp bias0 memcl iterilp}; do
  if [ $v = base ]; then cp /tmp/base.so sca_amd/lib/libsca_hip.so; else cp tools/_build/variants/$v.so sca_amd/lib/libsca_hip.so; fi
  SCA_BENCH_DETAIL=$O/c4drv_${v}_$rep.json timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_BENCH_DETAIL=$O/c4std_${v}_$rep.json timeout 300 python3 bench.py --steps 100 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_BENCH_DETAIL=$O/c2_${v}_$rep.json timeout 300 python3 bench.py --workload c2 --steps 300 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_BENCH_DETAIL=$O/c3auto_${v}_$rep.json timeout 300 python3 bench.py --workload c3 --nbr auto --steps 300 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  SCA_BENCH_DETAIL=$O/c5_${v}_$rep.json timeout 300 python3 bench.py --workload c5 --steps 300 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
done
done
cp /tmp/base.so sca_amd/lib/libsca_hip.so
python3 - $O <<'PY'
import json, os, sys, collections
O = sys.argv[1]
t = collections.defaultdict(dict)
for f in sorted(os.listdir(O)):
    if f.endswith('.json'):
        leg, v, rep = f[:-5].rsplit('_', 2)
        d = json.load(open(os.path.join(O, f)))
        t[leg].setdefault(v, []).append(d['ms_per_step'])
for leg in t:
    print(leg, '  '.join('%s %s' % (v, '/'.join('%.4f' % x for x in xs)) for v, xs in t[leg].items()))
PY
