#!/bin/bash
# round 6, second session: (1) the PCIe-inclusive rate (env_api.host_handover: INTEGRATION.md stub B), (2) c4 in the driver's form and at
# 100 steps with two host-side knobs: the fallback sweep inside the epilogue's launch at c4 too (SCA_ACTION_FB_MAX), subtrees of 1536
# beside the tracker (SCA_KD_WAVE_CAP)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_q
mkdir -p $O
cd $R
timeout 600 python3 bench.py --env-api-only c2,c3,c4 --steps 20 --warmup 5 > $O/env_api.json 2> $O/env_api.err
for rep in 1 2 3; do
for cfg in "base SCA_QUIET=1" "afb SCA_ACTION_FB_MAX=1000000" "cap1536 SCA_KD_WAVE_CAP=1536" "both SCA_ACTION_FB_MAX=1000000:SCA_KD_WAVE_CAP=1536"; do
  set -- $cfg
  name=$1
  envs=$(echo $2 | tr ':' ' ')
  env $envs SCA_BENCH_DETAIL=$O/drv_${name}_$rep.json timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
  env $envs SCA_BENCH_DETAIL=$O/std_${name}_$rep.json timeout 300 python3 bench.py --steps 100 --warmup 20 --no-extra --no-cpu-baseline > /dev/null 2> $O/err.txt
done
done
python3 - $O <<'PY'
import json, os, sys
O = sys.argv[1]
for f in sorted(os.listdir(O)):
    if f.endswith('.json') and f[:3] in ('drv', 'std'):
        d = json.load(open(os.path.join(O, f))); print('%-24s %.4f' % (f, d['ms_per_step']))
e = json.load(open(os.path.join(O, 'env_api.json')))['env_api']
for k, r in e.items():
    print(k, 'resident %.4f step %.4f step_actions %.4f host_handover %.4f (x%.2f)' % (r['resident_ms_per_step'], r['step']['ms_per_step'],
          r['step_actions']['ms_per_step'], r['host_handover']['ms_per_step'], r['host_handover']['over_resident']))
PY
