set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for T in CollaborativeStackingCart CollaborativeHammeringCart; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/w1_${T}_fetch -- python3 $R/bench.py --env $T --steps 20 --warmup 3 --no-cpu-baseline --variant-lib $R/human-robot-gym_amd/variant_w1.so > /tmp/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/w1_${T}_write -- python3 $R/bench.py --env $T --steps 20 --warmup 3 --no-cpu-baseline --variant-lib $R/human-robot-gym_amd/variant_w1.so > /tmp/w.log 2>&1
done
cd $R
python3 tools/profile_pmc_tasks_condense.py w1
rm -rf gpurun_out/w1_*_fetch gpurun_out/w1_*_write
