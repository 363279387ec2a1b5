#!/bin/bash
# rocprofv3 evidence for the cube kernel (BASELINE config 4 shape): kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes.
# Usage on the GPU box: bash tools/profile_capture_pp.sh <tag>
set -e
TAG=${1:-r02pp}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --env PickPlaceHumanCart --steps 40 --warmup 10 --preroll 300 --no-cpu-baseline > $R/gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --env PickPlaceHumanCart --steps 20 --warmup 3 --preroll 300 --no-cpu-baseline > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --env PickPlaceHumanCart --steps 20 --warmup 3 --preroll 300 --no-cpu-baseline > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/bench.py --env PickPlaceHumanCart --steps 20 --warmup 3 --preroll 300 --no-cpu-baseline > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python3 bench.py --env PickPlaceHumanCart --steps 100 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo done
