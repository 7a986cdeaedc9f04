#!/bin/bash
# round 5, last build: the GPU suite, smoke, both fuzzers at length, the two bench commands (-> profiles/r05_a_bench_*.json)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_p
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q 2>&1 | tail -2 | tee $O/final_gpu_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/final_gpu_suite.txt
timeout 900 python3 tools/fuzz_auto.py 91 400 2>&1 | tail -1 | tee $O/final_fuzz.txt
timeout 900 python3 tools/fuzz_track.py 91 300 2>&1 | tail -1 | tee -a $O/final_fuzz.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_c4_driver_cmd.json 2> $O/bench_driver.err
python3 bench.py > $O/bench_c4_default.json 2> $O/bench_default.err
python3 tools/gpu/exp_host.py > $O/host_enqueue.jsonl 2>> $O/err.txt
