cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p20
rocprofv3 --kernel-trace --output-format csv -d /tmp/p20 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-extras --steps 20 --warmup 5 > /dev/null 2>&1
python - "$(find /tmp/p20 -name 'k_kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if "table_kernel" in r["Kernel_Name"])
d = [(e - s) / 1e3 for s, e in rows]
g = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
print("launches", len(d))
print("durations:", " ".join("%.0f" % x for x in d))
print("gaps:", " ".join("%.1f" % x for x in g))
PY
