#!/bin/bash
# per-launch durations of a tick run (rocprofv3 kernel trace): tools/debug/tick_prof.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o t -- python /root/repo/bench.py --steps 200 --warmup 5 --no-extras > /tmp/pd_bench.json 2>/dev/null
python - "$(find /tmp/pd -name 't_kernel_stats.csv' | head -1)" "$*" <<PY
import csv, sys, json
print("env:", sys.argv[2], "| bench:", json.loads(open("/tmp/pd_bench.json").read().strip().splitlines()[-1])["ms_per_step"], "ms/step")
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    n = r["Name"]
    tag = "main" if "F1Op" in n else ("aux" if "GruOp" in n else n[:40])
    print("  %-40s calls %s avg %.1f max %.1f us" % (tag, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
