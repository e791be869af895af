#!/bin/bash
# partly filled ticks in their own halves order from N occupied stages on (measurement build): tools/debug/range_halves_sweep.sh
export BEATRICE_HIP_LIB=$PWD/build_variants/lib_meas.so
for n in 99 27 24 20 16 12 8 4 1 99 20 12; do
  BEATRICE_HIP_TICK_RANGE_HALVES=$n TAG="halves from $n" timeout 120 python tools/debug/short_run_rate.py 4 20 2>/dev/null | grep median
done
