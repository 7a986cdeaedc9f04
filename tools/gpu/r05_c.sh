#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
python3 bench.py --env-api-only c2,c3,c4 --steps 100 --warmup 10 > $O/env_api.json 2> $O/env_api.err
python3 -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -m gpu -q -x -k "env_api or closed_loop or end_to_end or resident" 2>&1 | tail -4
python3 - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05_c/env_api.json'))['env_api']
for k,v in d.items():
    print(k, {a:(round(b['ms_per_step'],4) if isinstance(b,dict) else (round(b,4) if isinstance(b,float) else b)) for a,b in v.items() if a!='workload'})
PY
