#!/bin/bash
# round 6 evidence set: bench lines, rocprofv3 kernel stats + traces per leg, device-side timelines (debug build), reconcile pair.
# Everything lands in gpurun_out/r06_p; tools/gpu/r06_collect.py (run in the build container) turns it into profiles/r05_*.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_p
mkdir -p $O
cd $R
SCA_BENCH_DETAIL=$O/bench_c4_driver_cmd_detail.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4_driver_cmd.json 2> $O/bench_driver.err
SCA_BENCH_DETAIL=$O/bench_c4_default_detail.json python3 bench.py > $O/bench_c4_default.json 2> $O/bench_default.err
SCA_BENCH_DETAIL=$O/r8gs_untraced_detail.json python3 bench.py --workload c4 --emulate-rank-of 8 --nbr grid --vpref straight --steps 100 --warmup 10 > $O/r8gs_untraced.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  SCA_BENCH_DETAIL=$O/${name}_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-extra > $O/$name.json 2> $O/$name.err
}
SCA_BENCH_DETAIL=$O/driver_cmd_traced_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/driver_cmd -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_traced.json 2> $O/driver_cmd_traced.err
prof c4_e2e --workload c4
prof c4_solver --workload c4 --vpref straight
prof c4_grid --workload c4 --nbr grid
prof c2_e2e --workload c2
prof c5_e2e --workload c5
prof c3 --workload c3
prof c3_auto --workload c3 --nbr auto
prof c3lp_auto --workload c3lp --nbr auto
prof heldout --workload heldout
SCA_BENCH_DETAIL=$O/c4_rank8_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_rank8 -- python3 $R/bench.py --workload c4 --emulate-rank-of 8 --steps 100 --warmup 10 > $O/c4_rank8.json 2> $O/c4_rank8.err
SCA_BENCH_DETAIL=$O/c4_rank8_grid_solver_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_rank8_grid_solver -- python3 $R/bench.py --workload c4 --emulate-rank-of 8 --nbr grid --vpref straight --steps 100 --warmup 10 > $O/c4_rank8_grid_solver.json 2> $O/c4_rank8_grid_solver.err
cd $R
# device-side timelines
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
for cfg in "c3 auto" "c3 kd" "c3 grid" "c3lp auto" "c2 kd" "c5 kd" "c4 kd" "heldout kd"; do
  set -- $cfg
  python3 tools/device_timeline.py $1 --nbr $2 --steps 40 -o $O/dtl_$1_$2.json > $O/dtl_$1_$2.txt 2>&1
done
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
find $O -name "*agent_info*" -delete
find $O -name "*_kernel_trace.csv" -size +12M -delete
du -sh $O
