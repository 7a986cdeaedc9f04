#!/bin/bash
# round 6: smoke(), then the fuzzers on the round's last build: SCA_NBR_AUTO against the kd-tree mode (the launch-free kd query as the
# host picks it, and FORCED for lists of any length with the back-off out of the way), the device tracker against the host tracker
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_fuzz
mkdir -p $O
cd $R
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
timeout 1500 python3 tools/fuzz_auto.py 601 200 > $O/fuzz_auto_default.txt 2>&1; echo "rc $?" >> $O/fuzz_auto_default.txt
SCA_AUTO_TAIL_MAX=1000000000 SCA_AUTO_BACKOFF_DIV=1 timeout 1500 python3 tools/fuzz_auto.py 602 120 > $O/fuzz_auto_forced_tail.txt 2>&1; echo "rc $?" >> $O/fuzz_auto_forced_tail.txt
SCA_AUTO_NO_TAIL=1 timeout 900 python3 tools/fuzz_auto.py 603 60 > $O/fuzz_auto_no_tail.txt 2>&1; echo "rc $?" >> $O/fuzz_auto_no_tail.txt
timeout 1500 python3 tools/fuzz_track.py 604 80 > $O/fuzz_track.txt 2>&1; echo "rc $?" >> $O/fuzz_track.txt
tail -2 $O/*.txt
