#!/bin/bash
# which re-plan form suits ONE RANK OF 8's shard (12 500 agents, ~12 200 plans per step at c4)?  The thresholds
# (SCA_TRK_*_MAX) were tuned on full-GPU shards, where 12 500 plans fall to k_replan_group<4>.  bench.py --emulate-rank-of 8 with the
# thresholds forced so that the shard's plans run as <4> (default), <16>, <32>, <64> (list + launch), k_track_group (64 lanes, fused with
# the decision), and one lane per plan.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_rank8_forms
mkdir -p $O
cd $R
run() {   # name, env assignments...
  name=$1; shift
  env "$@" SCA_BENCH_DETAIL=$O/$name.json python3 bench.py --emulate-rank-of 8 --steps 40 --warmup 10 > $O/$name.out 2> $O/$name.err
}
run quad4_default
run spec16 SCA_TRK_SPEC2_MAX=20000 SCA_TRACKER_NOGROUPFUSE=1
run spec32 SCA_TRK_SPEC3_MAX=20000 SCA_TRACKER_NOGROUPFUSE=1
run spec64 SCA_TRK_SPEC4_MAX=20000 SCA_TRACKER_NOGROUPFUSE=1
run track_group64 SCA_TRK_SPEC4_MAX=20000
run lane SCA_TRK_MID_MAX=0 SCA_TRK_SPEC2_MAX=0 SCA_TRK_SPEC3_MAX=0 SCA_TRK_SPEC4_MAX=0
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
print('%-16s %10s %12s %12s  %s' % ('form', 'ms/step', 're-plan ms', 'plans/step', 'kernel forms'))
for name in ('quad4_default', 'spec16', 'spec32', 'spec64', 'track_group64', 'lane'):
    try:
        d = json.load(open(os.path.join(O, name + '.json')))
    except Exception as e:
        print('%-16s failed: %s' % (name, e)); continue
    r = d['roofline']
    print('%-16s %10.4f %12.4f %12.1f  %s' % (name, d['ms_per_step'], r.get('kernel_ms', float('nan')) if r.get('unit_name') == 're-plan' else float('nan'),
                                              r.get('units_per_launch', 0) if r.get('unit_name') == 're-plan' else 0, '; '.join(d['config']['kernel_forms'])))
PY
