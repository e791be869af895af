#!/bin/bash
# time of the tick launch with one group of bodies left out (measurement build lib_meas.so, BEATRICE_HIP_TICK_DROP): the marginal
# time of each group inside the full launch.  groups (bit): 0 f1/fft/head/cond, 1 GRUs, 2 pitch convs, 3 inp/out, 4 tail, 5 blocks,
# 6 content-encoder convs, 7 up1/res1/up2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export BEATRICE_HIP_LIB=$ROOT/build_variants/${1:-lib_meas.so}
for drop in 0 1 2 4 8 16 32 64 128 0; do
  echo "drop $drop: $(BEATRICE_HIP_TICK_DROP=$drop python $ROOT/tools/debug/time_tick.py 2>/dev/null | grep -E 'TimeTickLaunch\((64|16)\)|loop' | sed -E 's/TimeTickLaunch\(([0-9]+)\): //; s/loop without drain: /loop /; s/ per tick//' | tr '\n' ' ')"
done
