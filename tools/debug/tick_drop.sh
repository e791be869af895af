#!/bin/bash
# tick duration with groups of bodies left out of the launch (BEATRICE_HIP_TICK_DROP bit mask, batch.hip:
# 1 per-stream small kernels, 2 GRUs, 4 pitch convs, 8 phone.out/wave.inp, 16 tail, 32 blocks, 64 phone convs, 128 wave convs)
# (BEATRICE_HIP_TICK_DROP exists in measurement builds only: first `tools/debug/build_variant.sh meas -DBEATRICE_HIP_MEASUREMENT_BUILD`;
#  this script then runs on build_variants/lib_meas.so)
[ -f "$(dirname "$0")/../../build_variants/lib_meas.so" ] && export BEATRICE_HIP_LIB="$(cd "$(dirname "$0")/../.." && pwd)/build_variants/lib_meas.so"
for D in ${DROPS:-0 1 2 3 15 16 32 64 128}; do
  BEATRICE_HIP_TICK_DROP=$D python bench.py --steps 300 --warmup 30 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('drop', $D, 'ms/tick', d['ms_per_step'])"
done
