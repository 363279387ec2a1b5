#!/bin/bash
# HBM traffic of the handover / lifting / inspection kernels: separate FETCH_SIZE and WRITE_SIZE passes (MI355X_MICROARCH.md: KiB units, FETCH_SIZE doubled on gfx950).
# Usage on the GPU box: bash tools/profile_capture_tasks_pmc.sh [tag]
set -e
TAG=${1:-r01t}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp
for T in ${TASKS:-HumanObjectInspectionCart HumanRobotHandoverCart RobotHumanHandoverCart CollaborativeLiftingCart CollaborativeStackingCart CollaborativeHammeringCart}; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${T}_fetch -- python3 $R/bench.py --env $T --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_${T}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${T}_write -- python3 $R/bench.py --env $T --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_${T}_write.log 2>&1
  echo "$T pmc done"
done
