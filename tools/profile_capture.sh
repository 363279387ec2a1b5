#!/bin/bash
# Capture the rocprofv3 evidence for profiles/: kernel trace + stats, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, three SQ passes).
# Every pass runs the default bench workload in its steady state (bench.py pre-rolls one horizon before the timed steps).
# Usage on the GPU box:  bash tools/profile_capture.sh <tag>     (writes gpurun_out/<tag>_*; condense with tools/profile_summarize.py <tag>)
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- $B --steps 100 --warmup 10 > $R/gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- $B --steps 20 --warmup 3 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- $B --steps 20 --warmup 3 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- $B --steps 20 --warmup 3 > $R/gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq2 -- $B --steps 20 --warmup 3 > $R/gpurun_out/${TAG}_sq2.log 2>&1
# lane utilisation and the FP64 instruction mix of the vector pipe ("show FP64 VALU utilisation alongside")
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq3 -- $B --steps 20 --warmup 3 > $R/gpurun_out/${TAG}_sq3.log 2>&1
cd $R
python3 bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --steps 200 --warmup 20 --shield OFF --no-cpu-baseline > gpurun_out/${TAG}_bench_off.json 2>> gpurun_out/${TAG}_bench.err
echo done
