#!/bin/bash
# Round 6: the tick as two kernels (dense + light, tick.hip.h kLightTypes) against the one-kernel launch, on ONE box, with the measurement build
# (tools/debug/build_variant.sh meas -DBEATRICE_HIP_MEASUREMENT_BUILD).  tools/debug/r06_split_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export BEATRICE_HIP_LIB=$ROOT/build_variants/lib_meas.so
rounds=${1:-2}
line() { python $ROOT/tools/debug/time_tick.py "$@" 2>/dev/null | grep -E 'TimeTickLaunch\((64|16)\)|loop' | sed -E 's/TimeTickLaunch\(([0-9]+)\): //; s/loop without drain: /loop /; s/ per tick//' | tr '\n' ' '; }
for r in $(seq $rounds); do
  for shape in "256 - 4" "256 - 2" "256 - 1" "1024 - 4" "64 - 4"; do
    echo "[$shape] one kernel : $(BEATRICE_HIP_TICK_ONE_KERNEL=1 line $shape)"
    echo "[$shape] two kernels: $(line $shape)"
  done
done
