#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite with K4 / a18 on the restated glibc (equality everywhere), the driver's bench command
# (compact last line), kernel stats of the c4 step (k_action with sincos / atan2 / pow2 on the constant tables)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -q -x --timeout 1200 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
SCA_BENCH_DETAIL=$O/bench_detail.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.out 2> $O/bench_driver.err
echo "bench rc $? bytes $(wc -c < $O/bench_driver.out)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c4 -- python3 $R/bench.py --workload c4 --steps 30 --warmup 10 --no-extra --no-cpu-baseline > $O/ks_c4.out 2> $O/ks_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c3 -- python3 $R/bench.py --workload c3 --nbr auto --steps 100 --warmup 10 --no-extra --no-cpu-baseline > $O/ks_c3.out 2> $O/ks_c3.err
cd $R
find $O -name "*agent_info*" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
