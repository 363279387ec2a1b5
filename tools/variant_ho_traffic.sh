#!/bin/bash
# HBM traffic + time of the handover kernel for alternative builds (HRG_LIB_PATH): bash tools/variant_ho_traffic.sh lib1.so lib2.so ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
for L in "$@"; do
  N=$(basename $L .so)
  V="--variant-lib $R/$L"
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/vho_${N}_fetch -- python3 $R/bench.py $V --env HumanRobotHandoverCart --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/vho_${N}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/vho_${N}_write -- python3 $R/bench.py $V --env HumanRobotHandoverCart --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/vho_${N}_write.log 2>&1
  cd $R
  python3 bench.py $V --env HumanRobotHandoverCart --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null > gpurun_out/vho_${N}_bench.json
  python3 bench.py $V --env RobotHumanHandoverCart --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null > gpurun_out/vho_${N}_bench_r2h.json
  python3 - <<PY
import csv, glob, json, os
v={}
for sub,cn in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    f=sorted(glob.glob("gpurun_out/vho_${N}_%s/*/*_counter_collection.csv" % sub), key=os.path.getmtime)[-1]
    x=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "hrg_step_kernel" in r["Kernel_Name"] and r["Counter_Name"]==cn]
    v[cn]=sum(x)/len(x)*1024/1e6
b=json.load(open("gpurun_out/vho_${N}_bench.json")); b2=json.load(open("gpurun_out/vho_${N}_bench_r2h.json"))
print("${N}: H2R kernel %.3f ms, R2H kernel %.3f ms, HBM MB/launch %.0f (fetch x2 %.0f + write %.0f)" % (b["roofline"]["kernel_ms"], b2["roofline"]["kernel_ms"], 2*v["FETCH_SIZE"]+v["WRITE_SIZE"], 2*v["FETCH_SIZE"], v["WRITE_SIZE"]))
PY
done
