#!/bin/bash
# rocprofv3 kernel-trace + stats and an un-profiled bench line for the collaboration tasks' kernel variants (box, handover, lifting).
# Usage on the GPU box: bash tools/profile_capture_tasks.sh [tag]      (writes gpurun_out/<tag>_<task>_*)
set -e
TAG=${1:-r02t}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
for T in ${TASKS:-HumanObjectInspectionCart HumanRobotHandoverCart RobotHumanHandoverCart CollaborativeLiftingCart CollaborativeStackingCart CollaborativeHammeringCart}; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${T}_trace -- python3 $R/bench.py --env $T --steps 40 --warmup 10 --preroll 200 --no-cpu-baseline > $R/gpurun_out/${TAG}_${T}_trace.log 2>&1
  cd $R
  python3 bench.py --env $T --steps 60 --warmup 10 --cpu-budget 6 > gpurun_out/${TAG}_${T}_bench.json 2> gpurun_out/${TAG}_${T}_bench.err
  echo "$T done"
done
