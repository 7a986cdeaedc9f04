#!/bin/bash
# round 5: sca_env_step leaves kd_stream unjoined (API_ENTER joins lazily) -- tests, then env_api c2/c3
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -x -k "env or auto or AUTO or params or abi" 2>&1 | tail -2
for rep in 1 2; do
python3 bench.py --env-api-only c2,c3 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline())
e=d.get("env_api", d)
for k,v in e.items():
    if isinstance(v,dict): print(k, round(v["step"]["ms_per_step"],4), round(v["resident_ms_per_step"],4), round(v["step_over_resident"],3))
'
done | tee $O/env_api_lazy.txt
