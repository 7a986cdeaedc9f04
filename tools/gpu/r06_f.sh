#!/bin/bash
# round 6: the whole -m gpu suite on the build with the tail form + k_action_fb, then device timelines
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_f
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "FAILED|passed|failed" $O/pytest.log | tail -20
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
for cfg in "c3 auto" "c3lp auto" "c3 kd" "c3 grid" "c2 kd" "c5 kd" "c4 kd" "heldout kd"; do
  set -- $cfg
  python3 tools/device_timeline.py $1 --nbr $2 --steps 40 -o $O/dtl_$1_$2.json > $O/dtl_$1_$2.txt 2>&1
done
SCA_AUTO_NO_TAIL=1 python3 tools/device_timeline.py c3 --nbr auto --steps 40 -o $O/dtl_c3_auto_notail.json > $O/dtl_c3_auto_notail.txt 2>&1
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
tail -16 $O/dtl_c3_auto.txt
