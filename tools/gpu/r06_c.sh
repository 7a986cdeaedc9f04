#!/bin/bash
# round 6, third GPU call: the whole -m gpu suite again (pow(x, 2) under cartesian2spherical's sqrt), then kernel stats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "FAILED|passed|failed" $O/pytest.log | tail -30
