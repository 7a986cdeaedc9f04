#!/bin/bash
# round 6, the evidence run on the round's last build: the whole -m gpu suite, then tools/gpu/r06_profiles.sh and r06_pmc.sh
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q --timeout 400 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "FAILED|passed|failed|Timeout" $O/pytest.log | tail -10
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
rm -rf $R/gpurun_out/r06_p $R/gpurun_out/r06_pmc
bash tools/gpu/r06_profiles.sh > $O/profiles.log 2>&1
bash tools/gpu/r06_pmc.sh > $O/pmc.log 2>&1
cat $R/gpurun_out/r06_p/bench_c4_driver_cmd.json
