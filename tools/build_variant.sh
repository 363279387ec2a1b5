#!/bin/bash
# Tuning aid: build another copy of the library with extra compiler flags -> human-robot-gym_amd/variant_<name>.so (git-ignored, travels with gpurun).
#   bash tools/build_variant.sh stamps -DHRG_STAMPS        (the diagnostic build tools/stamps.py loads)
# Time it with:  python bench.py --variant-lib human-robot-gym_amd/variant_<name>.so --no-cpu-baseline
set -e
N=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
C=$R/human-robot-gym_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Xarch_device -fapprox-func -mllvm -disable-machine-licm "$@" -o $R/human-robot-gym_amd/variant_$N.so $C/hrgym_hip.hip $C/hrgym_box.hip $C/hrgym_handover.hip $C/hrgym_lift.hip $C/hrgym_stack.hip $C/hrgym_hammer.hip $C/hrgym_hulls.hip
echo built variant_$N.so
