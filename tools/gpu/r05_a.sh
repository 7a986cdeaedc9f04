#!/bin/bash
# round 5, first measurement set: env_api legs, timelines (kernel traces) of c4 / c3 AUTO / one rank of 8, and the driver's command
# against the default one (why are the driver's 20 steps after 5 slower than 100 after 20?)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py --env-api-only c2,c3,c4 --steps 100 --warmup 10 > $O/env_api.json 2> $O/env_api.err
# the driver's command vs longer warm-ups / repeated legs (value leg only)
for cfg in "5 20" "20 20" "100 20" "5 100" "20 100"; do
  set -- $cfg
  python3 bench.py --warmup $1 --steps $2 --no-extra --no-cpu-baseline > $O/c4_w$1_s$2.json 2>> $O/c4_legs.err
done
# timelines
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl_c4 -- python3 $R/bench.py --workload c4 --steps 30 --warmup 10 --no-extra --no-cpu-baseline > $O/tl_c4.json 2> $O/tl_c4.err
rocprofv3 --kernel-trace --output-format csv -d $O/tl_c3auto -- python3 $R/bench.py --workload c3 --nbr auto --steps 30 --warmup 10 --no-extra --no-cpu-baseline > $O/tl_c3auto.json 2> $O/tl_c3auto.err
rocprofv3 --kernel-trace --output-format csv -d $O/tl_c3kd -- python3 $R/bench.py --workload c3 --steps 30 --warmup 10 --no-extra --no-cpu-baseline > $O/tl_c3kd.json 2> $O/tl_c3kd.err
rocprofv3 --kernel-trace --output-format csv -d $O/tl_c4r8 -- python3 $R/bench.py --workload c4 --emulate-rank-of 8 --steps 30 --warmup 10 > $O/tl_c4r8.json 2> $O/tl_c4r8.err
rocprofv3 --kernel-trace --output-format csv -d $O/tl_c4r8gs -- python3 $R/bench.py --workload c4 --emulate-rank-of 8 --nbr grid --vpref straight --steps 30 --warmup 10 > $O/tl_c4r8gs.json 2> $O/tl_c4r8gs.err
rocprofv3 --kernel-trace --output-format csv -d $O/tl_c2 -- python3 $R/bench.py --workload c2 --steps 30 --warmup 10 --no-extra --no-cpu-baseline > $O/tl_c2.json 2> $O/tl_c2.err
cd $R
# keep the traces small enough for the 64 MiB merge: only the kernel trace csv
find $O -name "*agent_info*" -delete
du -sh $O
ls $O
