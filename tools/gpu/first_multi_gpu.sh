#!/bin/bash
# The first lease with more than one GPU, in ONE command: no run with more than one RCCL rank has ever happened, so
# everything multi-GPU in this repository is correct by construction and tested with all ranks on one GPU.  This script turns the first
# real lease into evidence:
#   1. the two RCCL tests that skip on a 1-GPU box (tests/test_gpu_multirank.py::test_rccl_inside_the_library_two_ranks_bit_identical);
#   2. bench.py --gpus G for G in $SCA_FIRST_GPUS (default: 2 4 8, capped by the GPUs visible) with SCA_BENCH_BOTH_EXCHANGES=1 (the
#      all-gather through torch.distributed AND by the library's own RCCL communicator inside sca_run_steps), strong scaling, c4;
#   3. the cell-owner partition variant (--nbr grid --partition) at the same G;
#   4. one JSON (first_multi_gpu.json) with, per run: n_gpus, value, ms_per_step, rccl_ranks_seen, exchange, exchange_ms_measured,
#      the other exchange's line -- what replaces scale_model's MODELLED all-gather time.
# Usage:  bash tools/gpu/first_multi_gpu.sh [outdir]          (SCA_BENCH_SHARE_GPU=1: plumbing check with all ranks on GPU 0 over gloo --
#         what tests/test_gpu_bench.py::test_first_multi_gpu_script_plumbing runs; the RCCL tests then skip themselves)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=${1:-$R/gpurun_out/first_multi_gpu}
mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
SHARE=${SCA_BENCH_SHARE_GPU:-}
GLIST=${SCA_FIRST_GPUS:-"2 4 8"}
STEPS=${SCA_FIRST_STEPS:-20}
WARM=${SCA_FIRST_WARMUP:-5}
AGENTS=${SCA_FIRST_AGENTS:-}          # (empty: the workload's N = 100000)
echo "visible GPUs: $NGPU  share-one-GPU hook: ${SHARE:-no}  G in: $GLIST" | tee "$O/summary.txt"
python3 -m pytest tests/test_gpu_multirank.py -q -k rccl_inside_the_library -rs > "$O/rccl_tests.log" 2>&1
echo "rccl tests rc $? : $(tail -1 "$O/rccl_tests.log")" | tee -a "$O/summary.txt"
port=29700
for G in $GLIST; do
  if [ -z "$SHARE" ] && [ "$G" -gt "$NGPU" ]; then echo "G=$G skipped: only $NGPU GPUs" | tee -a "$O/summary.txt"; continue; fi
  for variant in allgather partition; do
    port=$((port + 1))
    extra=""; [ "$variant" = partition ] && extra="--nbr grid --partition"
    [ -n "$AGENTS" ] && extra="$extra --agents $AGENTS"
    SCA_BENCH_BOTH_EXCHANGES=1 SCA_BENCH_DETAIL="$O/g${G}_${variant}_detail.json" MASTER_PORT=$port timeout 1800 \
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$G" --master-addr 127.0.0.1 --master-port $port bench.py \
      --gpus "$G" --steps "$STEPS" --warmup "$WARM" $extra > "$O/g${G}_${variant}.out" 2> "$O/g${G}_${variant}.err"
    echo "G=$G $variant rc $? : $(grep '^{' "$O/g${G}_${variant}.out" | tail -1 | cut -c1-160)" | tee -a "$O/summary.txt"
  done
done
python3 - "$O" <<'PY'
import glob, json, os, sys
O = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(O, 'g*_detail.json'))):
    d = json.load(open(f))
    pg = d.get('process_group') or {}
    rows.append({'file': os.path.basename(f), 'n_gpus': d['n_gpus'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'scaling': d['scaling'],
                 'agents': d['config']['agents'], 'rccl_ranks_seen': d.get('rccl_ranks_seen'), 'backend': pg.get('backend'), 'exchange': pg.get('exchange'),
                 'exchange_ms_measured': pg.get('exchange_ms_measured'), 'ranks_sharing_gpu0_test_hook': pg.get('ranks_sharing_gpu0_test_hook'),
                 'other_exchange': d.get('other_exchange')})
json.dump({'runs': rows, 'note': 'speed-up = value / the 1-GPU value of the same bench.py (BENCH_rNN.json); exchange_ms_measured replaces '
                                 'scale_model.allgather_ms_assumed'}, open(os.path.join(O, 'first_multi_gpu.json'), 'w'), indent=1)
print(json.dumps(rows, indent=1))
PY
