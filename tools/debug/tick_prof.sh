#!/bin/bash
# per-launch durations of a tick run (rocprofv3 kernel trace):
#   tools/debug/tick_prof.sh [env assignments...] [-- bench.py arguments]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
timeout 240 env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o t -- python /root/repo/bench.py --steps 200 --warmup 5 --no-extras "$@" > /tmp/pd_bench.json 2>/dev/null
f="$(find /tmp/pd -name 't_kernel_stats.csv' | head -1)"
[ -n "$f" ] || { echo "no kernel stats produced"; exit 1; }
python - "$f" "${envs[*]} $*" <<PY
import csv, sys, json
print("env/args:", sys.argv[2], "| bench:", json.loads(open("/tmp/pd_bench.json").read().strip().splitlines()[-1])["ms_per_step"], "ms/step")
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    n = r["Name"]
    tag = "tick launch" if "F1Op" in n else n[:48]
    print("  %-48s calls %s avg %.1f max %.1f us" % (tag, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
