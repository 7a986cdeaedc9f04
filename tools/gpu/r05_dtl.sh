#!/bin/bash
# device-side timelines on the round's last build (the end stamps of small launches are taken by every wavefront since the first set)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_p
mkdir -p $O
cd $R
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
for cfg in "c3 auto" "c3 kd" "c3 grid" "c3lp auto" "c2 kd" "c5 kd" "c4 kd" "heldout kd"; do
  set -- $cfg
  python3 tools/device_timeline.py $1 --nbr $2 --steps 40 -o $O/dtl_$1_$2.json > $O/dtl_$1_$2.txt 2>&1
done
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
head -12 $O/dtl_c2_kd.txt; head -14 $O/dtl_c3_auto.txt
