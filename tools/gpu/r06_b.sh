#!/bin/bash
# round 6, second GPU call: the whole -m gpu suite (no -x), the device-wide barrier against dependent dispatches (tools/bench/grid_barrier.hip)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
timeout 120 tools/_build/grid_barrier > $O/grid_barrier.txt 2>&1
echo "grid_barrier rc $?" >> $O/grid_barrier.txt
cat $O/grid_barrier.txt
timeout 3000 python3 -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
