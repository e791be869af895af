#!/bin/bash
# timeline of a short tick run (fill + drain), last of three repetitions: per launch its start (us after the first), duration and the idle gap before it;
# and the host's time per feed call (printed by fill_drain.py with FD_HOSTTIMES=1):  tools/debug/fill_drain_timeline.sh [steps] [env...]
steps=${1:-20}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfd
env FD_HOSTTIMES=1 "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfd -o k -- python $GRAFT_REPO_ROOT/tools/debug/fill_drain.py $steps 2>/dev/null | grep "^host"
python - "$(find /tmp/pfd -name 'k_kernel_trace.csv' | head -1)" $steps "$*" <<'PY'
import csv, sys
steps = int(sys.argv[2])
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if "table_kernel" in r["Kernel_Name"])
rows = rows[-(steps + 27):]
t0 = rows[0][0]
print(sys.argv[3], "| %d launches, sum %.0f us, span %.0f us" % (len(rows), sum(e - s for s, e in rows) / 1e3, (rows[-1][1] - t0) / 1e3))
prev = None
for i, (s, e) in enumerate(rows):
    print("  launch %2d  start %7.1f  dur %6.1f  gap %5.1f" % (i, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
