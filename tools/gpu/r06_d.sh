#!/bin/bash
# round 6, fourth GPU call: two new tests, rank-of-8 re-plan forms, the AUTO back-off threshold (1/8 vs 1/2 vs all of the shard) on c4,
# device-side timelines (SCA_TIMELINE build) of c3 AUTO / c3 kd / c2 / c4
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_d
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_partition.py::test_partition_and_per_agent_attributes_in_either_order tests/test_gpu_bench.py::test_first_multi_gpu_script_plumbing -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash tools/gpu/r06_rank8_forms.sh > $O/rank8_forms.txt 2>&1
cat $O/rank8_forms.txt
# AUTO back-off: the grid query lists an agent for the kd query when it has more than 16 objects in range or equal rounded distances; AUTO
# falls back to the plain kd pass for 256 passes once more than 1 / DIV of the shard is listed
for div in 8 2 1; do
  SCA_AUTO_BACKOFF_DIV=$div SCA_BENCH_DETAIL=$O/c4_auto_straight_div$div.json python3 bench.py --nbr auto --vpref straight --steps 60 --warmup 20 --no-extra --no-cpu-baseline > $O/c4_auto_straight_div$div.out 2> $O/c4_auto_straight_div$div.err
  SCA_AUTO_BACKOFF_DIV=$div SCA_BENCH_DETAIL=$O/c4_auto_value_div$div.json python3 bench.py --nbr auto --steps 40 --warmup 20 --no-extra --no-cpu-baseline > $O/c4_auto_value_div$div.out 2> $O/c4_auto_value_div$div.err
done
SCA_BENCH_DETAIL=$O/c4_kd_straight.json python3 bench.py --nbr kd --vpref straight --steps 60 --warmup 20 --no-extra --no-cpu-baseline > $O/c4_kd_straight.out 2> $O/c4_kd_straight.err
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.startswith('c4_') and f.endswith('.json'):
        d = json.load(open(os.path.join(O, f)))
        print('%-34s ms/step %.4f  value %.4g' % (f, d['ms_per_step'], d['value']))
PY
# device timelines
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
for cfg in "c3 auto" "c3 kd" "c2 kd" "c4 kd"; do
  set -- $cfg
  python3 tools/device_timeline.py $1 --nbr $2 --steps 40 -o $O/dtl_$1_$2.json > $O/dtl_$1_$2.txt 2>&1
done
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
head -20 $O/dtl_c3_auto.txt
