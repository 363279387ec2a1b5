#!/bin/bash
# One call on the GPU box: every capture of a round, condensed there (the raw rocprofv3 CSVs exceed gpurun's 64 MiB return limit) -> gpurun_out/summ/<tag>*.
#   bash tools/profile_capture_all.sh r03        then locally: cp gpurun_out/summ/r03* profiles/
# headline (trace + FETCH/WRITE + three SQ passes + bench lines), PickPlace 8192 (r03pp), the six task variants (r03t: trace, bench line, FETCH/WRITE),
# the mixed batch, the hull geometry, the Cartesian front-end, and the soak runs.
set -e
TAG=${1:-r03}
PART=${2:-all}   # A = headline, PickPlace, mixed, geometry; B = the task variants and the soak runs (each fits one gpurun call)
R=${GRAFT_REPO_ROOT:-$PWD}
S=$R/gpurun_out/summ
mkdir -p $S
cd $R
say() { echo "[$(date +%H:%M:%S)] $*"; }
if [[ $PART == all || $PART == A ]]; then
say headline;   bash tools/profile_capture.sh $TAG > gpurun_out/${TAG}_capture.log 2>&1
python3 tools/profile_summarize.py $TAG > gpurun_out/${TAG}_summarize.log 2>&1
cp gpurun_out/${TAG}_trace/*/*_kernel_stats.csv $S/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3
say pickplace;  bash tools/profile_capture_pp.sh ${TAG}pp > gpurun_out/${TAG}pp_capture.log 2>&1
python3 tools/profile_summarize.py ${TAG}pp '`python3 bench.py --env PickPlaceHumanCart --steps N --warmup W --preroll 300 --no-cpu-baseline` (1 x MI355X, 8192 envs, SSM)' > gpurun_out/${TAG}pp_summarize.log 2>&1
cp gpurun_out/${TAG}pp_trace/*/*_kernel_stats.csv $S/${TAG}pp_kernel_stats.csv
rm -rf gpurun_out/${TAG}pp_trace gpurun_out/${TAG}pp_fetch gpurun_out/${TAG}pp_write gpurun_out/${TAG}pp_sq
fi
if [[ $PART == all || $PART == B ]]; then
say tasks;      bash tools/profile_capture_tasks.sh ${TAG}t > gpurun_out/${TAG}t_capture.log 2>&1
say tasks-pmc;  bash tools/profile_capture_tasks_pmc.sh ${TAG}t > gpurun_out/${TAG}t_capture_pmc.log 2>&1
python3 tools/profile_pmc_tasks_condense.py ${TAG}t > gpurun_out/${TAG}t_condense.log 2>&1
python3 tools/profile_summarize_tasks.py ${TAG}t > gpurun_out/${TAG}t_summarize.log 2>&1
rm -rf gpurun_out/${TAG}t_*_trace gpurun_out/${TAG}t_*_fetch gpurun_out/${TAG}t_*_write
fi
if [[ $PART == all || $PART == A ]]; then
say mixed;      cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_mixed_trace -- python3 $R/bench.py --env mixed --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${TAG}_mixed_trace.log 2>&1
cd $R
python3 tools/profile_kernel_stats_md.py gpurun_out/${TAG}_mixed_trace > $S/${TAG}_mixed_kernel_stats.md 2> gpurun_out/${TAG}_mixed_stats.err || cp gpurun_out/${TAG}_mixed_trace/*/*_kernel_stats.csv $S/${TAG}_mixed_kernel_stats.csv
rm -rf gpurun_out/${TAG}_mixed_trace
python3 bench.py --env mixed --steps 200 --warmup 20 > $S/${TAG}_bench_mixed.json 2> gpurun_out/${TAG}_bench_mixed.err
say geometry;   python3 bench.py --robot-geometry hull --no-cpu-baseline > $S/${TAG}_bench_hull.json 2> gpurun_out/${TAG}_bench_hull.err
python3 bench.py --robot-geometry hull --shield OFF --no-cpu-baseline > $S/${TAG}_bench_hull_off.json 2>> gpurun_out/${TAG}_bench_hull.err
python3 bench.py --collision-prevention --no-cpu-baseline > $S/${TAG}_bench_cp.json 2>> gpurun_out/${TAG}_bench_hull.err
python3 bench.py --env PickPlaceHumanCart --no-cpu-baseline > $S/${TAG}_bench_pickplace.json 2> gpurun_out/${TAG}_bench_pp.err
python3 bench.py --env PickPlaceHumanCart --ik --no-cpu-baseline > $S/${TAG}_bench_pickplace_ik.json 2>> gpurun_out/${TAG}_bench_pp.err
fi
if [[ $PART == all || $PART == B ]]; then
say soak;       python3 tools/soak_tasks.py > $S/${TAG}_soak_tasks.log 2>&1
python3 tools/soak_hammering.py 1500 random > $S/${TAG}_soak_hammering_random.log 2>&1
python3 tools/soak_hammering.py 600 still > $S/${TAG}_soak_hammering_still.log 2>&1
fi
cp -n profiles/${TAG}* $S/ 2>/dev/null || true   # (what the summarisers wrote into profiles/; never over a file this run produced)
(nproc; lscpu | head -20; cat /sys/fs/cgroup/cpu.max 2>/dev/null; uptime) > $S/${TAG}_host.txt 2>&1
say done; ls $S
