#!/bin/bash
# round 5: VALU calibration, the held-out speculation scene, multirank 4 / 8 on one GPU, and the rank-of-8 grid leg untraced vs traced
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
tools/_build/valu_calib > $O/valu_calib.json 2> $O/valu_calib.err
rocm-smi --showclocks > $O/clocks.txt 2>&1
python3 -m pytest tests/test_gpu_multirank.py -m gpu -q -x -k "one_gpu_match" 2>&1 | tail -4
python3 -m pytest tests/test_gpu_tracker.py tests/test_gpu_bench.py -m gpu -q -x -k "kats or env_api or small_swarm" 2>&1 | tail -4
# untraced / traced: one rank of eight, grid, solver alone
python3 bench.py --workload c4 --emulate-rank-of 8 --nbr grid --vpref straight --steps 100 --warmup 10 > $O/r8gs_untraced.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r8gs_traced -- python3 $R/bench.py --workload c4 --emulate-rank-of 8 --nbr grid --vpref straight --steps 100 --warmup 10 > $O/r8gs_traced.json 2>> $O/err.txt
cd $R
# the extra legs incl. the held-out scene (the whole default command, as the driver runs it but with the default K / W)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
find $O -name "*agent_info*" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
