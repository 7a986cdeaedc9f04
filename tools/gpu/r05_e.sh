#!/bin/bash
# round 5: device-side (un-perturbed) timelines from the SCA_TIMELINE debug build, VALU calibration (select pair fixed), extra legs
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_e
mkdir -p $O
cd $R
tools/_build/valu_calib > $O/valu_calib.json 2> $O/valu_calib.err
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
for cfg in "c3 auto" "c3 kd" "c3 grid" "c3lp auto" "c2 kd" "c5 kd" "c4 kd"; do
  set -- $cfg
  python3 tools/device_timeline.py $1 --nbr $2 --steps 40 -o $O/tl_$1_$2.json > $O/tl_$1_$2.txt 2>&1
done
python3 tools/device_timeline.py c4 --nbr grid --straight --steps 40 -o $O/tl_c4_grid_straight.json > $O/tl_c4_grid_straight.txt 2>&1
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
python3 bench.py --no-cpu-baseline --no-weak-model > $O/bench_default.json 2> $O/bench_default.err
head -30 $O/tl_c3_auto.txt
