export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cp $R/sca_amd/lib/libsca_hip.so /tmp/full.so
for e in 1 2 3 full; do
  if [ $e = full ]; then cp /tmp/full.so $R/sca_amd/lib/libsca_hip.so; else cp $R/scratch/exp$e/libsca_hip.so $R/sca_amd/lib/libsca_hip.so; fi
  mkdir -p $R/gpurun_out/exp_$e; cd /tmp
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/exp_$e -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  cd $R; python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/exp_$e/*/*counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_solve(" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$e", {k: round(sum(v)/len(v)/100000,1) for k,v in agg.items() if k!="SQ_WAVES"})
PY
done
