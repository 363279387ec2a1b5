#!/bin/bash
# A/B: HBM traffic (FETCH_SIZE, WRITE_SIZE passes) + kernel time of an alternative library build.  Usage: bash tools/variant_traffic2.sh <lib.so> <tag>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
V="--variant-lib $1"
T=$2
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/bench.py $V --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/bench.py $V --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${T}_write.log 2>&1
cd $R
python3 bench.py $V --no-cpu-baseline | cut -c1-160
python3 - <<PY
import csv, glob
for sub in ("fetch", "write"):
    f = sorted(glob.glob("$R/gpurun_out/${T}_%s/*/*_counter_collection.csv" % sub))[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "hrg_step_kernel" in r["Kernel_Name"]]
    print("$T", sub, "MB per launch", sum(v) / len(v) * 1024 / 1e6)
PY
