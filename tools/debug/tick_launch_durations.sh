cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o k -- python $GRAFT_REPO_ROOT/bench.py --no-extras --steps 300 --warmup 20 > /dev/null 2>&1
python - "$(find /tmp/pt -name 'k_kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "table_kernel" in r["Kernel_Name"])
ref = d[(3 * len(d)) // 4]
full = [x for x in d if x >= 0.9 * ref]
print("tick launches %d: mean of all %.2f us; full ticks (>= 0.9 x third quartile) %d: mean %.2f us, median %.2f us, max %.2f us; partly filled %d: mean %.2f us"
      % (len(d), sum(d) / len(d) / 1e3, len(full), sum(full) / len(full) / 1e3, full[len(full) // 2] / 1e3, d[-1] / 1e3,
         len(d) - len(full), (sum(d) - sum(full)) / max(1, len(d) - len(full)) / 1e3))
PY
