#!/bin/bash
# per-launch durations of a short tick run (fill + drain), last of three repetitions:  tools/debug/fill_drain.sh [steps] [env...]
steps=${1:-20}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfd
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/pfd -o k -- python $GRAFT_REPO_ROOT/tools/debug/fill_drain.py $steps > /dev/null 2>&1
python - "$(find /tmp/pfd -name 'k_kernel_trace.csv' | head -1)" $steps "$*" <<'PY'
import csv, sys
steps = int(sys.argv[2])
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if "table_kernel" in r["Kernel_Name"])
rows = rows[-(steps + 27):]
d = [(e - s) / 1e3 for s, e in rows]
print(sys.argv[3], "| %d launches, sum %.0f us, span %.0f us:" % (len(d), sum(d), (rows[-1][1] - rows[0][0]) / 1e3), " ".join("%.0f" % x for x in d))
PY
