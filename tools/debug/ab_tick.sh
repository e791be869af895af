#!/bin/bash
# A/B of the tick launch on ONE box: tools/debug/ab_tick.sh <rounds> <lib or "product"> ...  (boxes of the pool differ by +-3 %,
# so two builds are only comparable inside one gpurun call); prints the settled TimeTickLaunch figures of every run.
# extra arguments for time_tick.py through AB_ARGS
rounds=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset BEATRICE_HIP_LIB; else export BEATRICE_HIP_LIB=$ROOT/build_variants/$lib; fi
    echo "$lib: $(python $ROOT/tools/debug/time_tick.py $AB_ARGS 2>/dev/null | grep -E 'TimeTickLaunch\((64|16)\)|loop' | sed -E 's/TimeTickLaunch\(([0-9]+)\): //; s/loop without drain: /loop /; s/ per tick//' | tr '\n' ' ')"
  done
done
