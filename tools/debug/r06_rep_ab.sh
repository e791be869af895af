#!/bin/bash
# Round 6: the tiny bodies of the tick (f1, fft, cond) with several workgroup indices per workgroup (fuse.hip.h kRepeat), and where they sit in
# dispatch order.  Measurement build (tools/debug/build_variant.sh meas -DBEATRICE_HIP_MEASUREMENT_BUILD).  tools/debug/r06_rep_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export BEATRICE_HIP_LIB=$ROOT/build_variants/lib_meas.so
rounds=${1:-2}
line() { python $ROOT/tools/debug/time_tick.py "$@" 2>/dev/null | grep -E 'TimeTickLaunch\((64|16)\)|loop' | sed -E 's/TimeTickLaunch\(([0-9]+)\): //; s/loop without drain: /loop /; s/ per tick//' | tr '\n' ' '; }
for r in $(seq $rounds); do
  for shape in "256 - 4" "256 - 1" "1024 - 4"; do
    for rep in 1 2 4 8; do echo "[$shape] rep $rep at end : $(BEATRICE_HIP_TICK_REP=$rep line $shape)"; done
    for at in 1 2; do echo "[$shape] rep 4 at $at   : $(BEATRICE_HIP_TICK_REP=4 BEATRICE_HIP_TICK_TINY_AT=$at line $shape)"; done
    echo "[$shape] rep 8 at 2   : $(BEATRICE_HIP_TICK_REP=8 BEATRICE_HIP_TICK_TINY_AT=2 line $shape)"
  done
done
