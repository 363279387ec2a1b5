#!/bin/bash
# tuning experiment: HBM write/fetch counters per launch for prebuilt library variants
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pmc_$v_$c
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${v}_$c -- python3 $R/bench.py --variant-lib $R/human-robot-gym_amd/variant_$v.so --steps 30 --warmup 100 --no-cpu-baseline > /dev/null 2>&1)
    python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pmc_${v}_$c/*/*_counter_collection.csv")[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "hrg_step_kernel" in r["Kernel_Name"]]
print("$v $c per launch: %.1f MB" % (sum(vals[-30:]) / 30 * 1024 / 1e6))
PY
  done
done
