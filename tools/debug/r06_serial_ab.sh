#!/bin/bash
# Round 6: does the light class ALONE run faster at four workgroups per compute unit?  The two kernels of a tick one after the other on the
# batch's stream (BEATRICE_HIP_TICK_LIGHT_SERIAL=1: dense then light; 2: light then dense) against the one-kernel launch and the concurrent form.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export BEATRICE_HIP_LIB=$ROOT/build_variants/lib_meas.so
rounds=${1:-2}
line() { python $ROOT/tools/debug/time_tick.py "$@" 2>/dev/null | grep -E 'TimeTickLaunch\((64|16)\)|loop' | sed -E 's/TimeTickLaunch\(([0-9]+)\): //; s/loop without drain: /loop /; s/ per tick//' | tr '\n' ' '; }
for r in $(seq $rounds); do
  for shape in "256 - 4" "1024 - 4"; do
    echo "[$shape] one kernel        : $(BEATRICE_HIP_TICK_ONE_KERNEL=1 line $shape)"
    echo "[$shape] dense, then light : $(BEATRICE_HIP_TICK_LIGHT_SERIAL=1 line $shape)"
    echo "[$shape] light, then dense : $(BEATRICE_HIP_TICK_LIGHT_SERIAL=2 line $shape)"
    echo "[$shape] side by side      : $(line $shape)"
  done
done
