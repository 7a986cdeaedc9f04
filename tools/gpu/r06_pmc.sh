#!/bin/bash
# round 6 PMC passes (each counter group in a run of its own, --kernel-trace only beside --pmc): the c4 value leg and c3 (kd and AUTO)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $O/raw_$name -- python3 $R/bench.py "$@" --steps 8 --warmup 12 --no-cpu-baseline --no-extra --no-env-api > $O/$name.json 2> $O/$name.err
  python3 $R/tools/pmc_summary.py $O/raw_$name $O/$name.csv >> $O/summary.log 2>&1
  rm -rf $O/raw_$name
}
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
SQL="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
for leg in "c4_e2e --workload c4" "c3 --workload c3" "c3_auto --workload c3 --nbr auto"; do
  set -- $leg
  name=$1; shift
  if [ "$name" = "c4_e2e" ]; then pass r06_a_${name}_pmc_sq_counters "$SQ" "$@"; else pass r06_a_${name}_pmc_sq_counters "$SQL" "$@"; fi
  pass r06_a_${name}_pmc_fetch_size "FETCH_SIZE" "$@"
  pass r06_a_${name}_pmc_write_size "WRITE_SIZE" "$@"
done
ls -la $O; tail -3 $O/summary.log
