#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_f
mkdir -p $O
cd $R
python3 tools/gpu/exp_host.py 2>&1 | head -4
python3 -m pytest tests/test_gpu_auto.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -x -k "auto or AUTO" 2>&1 | tail -3
cp sca_amd/lib/libsca_hip.so /tmp/libsca_hip_product.so
SCA_BUILD_DEFS=-DSCA_TIMELINE python3 -m sca_amd.build > $O/build_tl.log 2>&1
for cfg in "c3 auto" "c3 kd" "c3 grid" "c2 kd" "c5 kd" "c4 kd"; do
  set -- $cfg
  python3 tools/device_timeline.py $1 --nbr $2 --steps 40 -o $O/tl_$1_$2.json > $O/tl_$1_$2.txt 2>&1
done
cp /tmp/libsca_hip_product.so sca_amd/lib/libsca_hip.so
cat $O/tl_c3_auto.txt
