#!/bin/bash
# per-workgroup timeline of a full tick: tools/debug/tick_trace_run.sh [DROP mask] [extra bench.py arguments...]
# (BEATRICE_HIP_TICK_DROP exists in measurement builds only: first `tools/debug/build_variant.sh meas -DBEATRICE_HIP_MEASUREMENT_BUILD`;
#  this script then runs on build_variants/lib_meas.so)
[ -f "$(dirname "$0")/../../build_variants/lib_meas.so" ] && export BEATRICE_HIP_LIB="$(cd "$(dirname "$0")/../.." && pwd)/build_variants/lib_meas.so"
export BEATRICE_HIP_TICK_DROP=${1:-0}
shift
BEATRICE_HIP_TICK_TRACE=/tmp/tick_trace.txt python bench.py --steps 100 --warmup 30 --no-extras "$@" > /dev/null 2>&1
python tools/debug/tick_trace.py /tmp/tick_trace.txt
