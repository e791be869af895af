#!/bin/bash
# Only the PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE, MFMA busy; separate runs) + their summary, for a change of the
# tick launch's sources after the round's full profile run: tools/pmc_restamp.sh <tag>  -> gpurun_out/<tag>/pmc_traffic.json + per-kernel means
set -u
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf "$OUT/pmc_$C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python "$ROOT/bench.py" --no-extras --steps 200 --warmup 5 > /dev/null 2>&1
  CSV=$(find "$OUT/pmc_$C" -name 'pmc_counter_collection.csv' | head -1)
  mkdir -p "$OUT/pmc_r1_$C"
  [ -n "$CSV" ] && cp "$CSV" "$OUT/pmc_r1_$C/pmc_counter_collection.csv"
  rm -rf "$OUT/pmc_$C"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" "$OUT/pmc_traffic.json"
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  python - "$OUT/pmc_r1_$C/pmc_counter_collection.csv" "$C" <<'PY'
import collections, csv, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
w = csv.writer(open(sys.argv[1].replace("pmc_counter_collection.csv", "per_kernel_mean.csv"), "w"))
w.writerow(["kernel", "launches", "mean_%s%s" % (sys.argv[2], "_KiB" if sys.argv[2].endswith("SIZE") else ""), "median", "max"])
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    w.writerow([k[:160], len(v), "%.2f" % (sum(v) / len(v)), "%.2f" % v2[len(v2) // 2], "%.2f" % v2[-1]])
PY
  rm -f "$OUT/pmc_r1_$C/pmc_counter_collection.csv"
done
