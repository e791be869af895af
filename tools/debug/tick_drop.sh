#!/bin/bash
# main tick launch duration with groups of bodies left out (see BEATRICE_HIP_TICK_DROP in batch.hip)
cd /tmp && export TMPDIR=/tmp
for D in ${DROPS:-0 62 61 59 55 47 31}; do
  for F in "" 1; do
  if [ -z "$F" ]; then export BEATRICE_HIP_TICK_XCD=1; else unset BEATRICE_HIP_TICK_XCD; fi
  rm -rf /tmp/pd; BEATRICE_HIP_TICK_DROP=$D rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o t -- python /root/repo/bench.py --steps 150 --warmup 5 --no-extras > /dev/null 2>&1
  S=$(find /tmp/pd -name "t_kernel_stats.csv" | head -1)
  python - "$S" "$D" "$F" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "F1Op" in r["Name"]: print("drop", sys.argv[2], "flat" if sys.argv[3] else "xcd ", "main launch max us %.1f avg %.1f" % (float(r["MaxNs"])/1e3, float(r["AverageNs"])/1e3))
PY
  done
done
