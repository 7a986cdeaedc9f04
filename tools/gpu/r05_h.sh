#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
for rep in $(seq 1 14); do
  python3 bench.py --workload c3 --nbr auto --steps 200 --warmup 50 --no-extra --no-cpu-baseline --no-env-api 2>/dev/null > $O/c3rep_$rep.json
  python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[1][-12:], round(d["ms_per_step"],4), d.get("auto"), d["config"].get("kernel_forms"), d.get("kd_build_ms"), {k:round(v,4) for k,v in d.get("roofline",{}).get("candidates_ms",{}).items()} if d.get("roofline") else None)
' $O/c3rep_$rep.json
done
