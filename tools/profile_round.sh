#!/bin/bash
# Measurement recipe of one round (run on a GPU box from the repo root, e.g. through gpurun):
#   tools/profile_round.sh r01_e
# (two sets of profiled passes: the headline configuration -- tick pipelining, whose dominant kernel is the tick launch --
# and the chain in order on one stream, --pipeline off, where a kernel's duration is that of the kernel alone)
# writes under gpurun_out/<tag>/: the bench line, the rocprofv3 kernel-trace/stats summary of the same
# command, and three separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) reduced to
# per-kernel means.  Copy what
# is to be judged into profiles/.
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

timeout 900 python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 400 "$OUT/bench.err"
# the driver's own flags, five times on this box (a 20-step run is 5 ms of launches and follows the box's clocks: VERDICT r05 item 3 asks for the median)
for i in 1 2 3 4 5; do timeout 300 python "$ROOT/bench.py" --steps 20 --warmup 5 --no-extras > "$OUT/bench_driver_flags_steps20_run$i.json" 2> /dev/null; done
timeout 600 python "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_driver_flags_steps20.json" 2> /dev/null
timeout 300 python "$ROOT/bench.py" --config 3 --no-extras > "$OUT/bench_config3.json" 2> /dev/null
timeout 300 python "$ROOT/bench.py" --config 4 --no-extras > "$OUT/bench_config4.json" 2> /dev/null
timeout 300 python "$ROOT/bench.py" --streams 1024 --no-extras > "$OUT/bench_B1024.json" 2> /dev/null

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- \
  python "$ROOT/bench.py" --no-extras --steps 300 --warmup 20 > "$OUT/bench_under_rocprof.json" 2> /dev/null
STATS=$(find "$OUT/prof" -name 'k_kernel_stats.csv' | head -1)
[ -n "$STATS" ] && cp "$STATS" "$OUT/kernel_stats_B256_tick.csv"
# the stats average of the tick launch mixes full ticks with the partly filled ones of fill and drain: split them
python - "$(find "$OUT/prof" -name 'k_kernel_trace.csv' | head -1)" > "$OUT/tick_launch_durations.txt" <<'PY'
import csv, sys
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "table_kernel" in r["Kernel_Name"])
ref = d[(3 * len(d)) // 4]   # third quartile: a full tick (more than half of the launches are full; an outlier does not move it)
full = [x for x in d if x >= 0.9 * ref]
print("tick launches %d: mean of all %.2f us; full ticks (>= 0.9 x third quartile) %d: mean %.2f us, median %.2f us, max %.2f us; partly filled %d: mean %.2f us"
      % (len(d), sum(d) / len(d) / 1e3, len(full), sum(full) / len(full) / 1e3, full[len(full) // 2] / 1e3, d[-1] / 1e3,
         len(d) - len(full), (sum(d) - sum(full)) / max(1, len(d) - len(full)) / 1e3))
PY
rm -rf "$OUT/prof"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- \
  python "$ROOT/bench.py" --no-extras --pipeline off --steps 200 --warmup 20 > /dev/null 2>&1
STATS=$(find "$OUT/prof" -name 'k_kernel_stats.csv' | head -1)
[ -n "$STATS" ] && cp "$STATS" "$OUT/kernel_stats_B256_in_order.csv"
rm -rf "$OUT/prof"
# configs[1]: the 1-stream C-ABI, eager launches (rocprofv3 does not survive the per-call graphs)
( cd "$ROOT" && bash tools/debug/b1_prof.sh "$TAG/b1" > /dev/null 2>&1 )

for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- \
    python "$ROOT/bench.py" --no-extras --steps 200 --warmup 5 > /dev/null 2>&1
  CSV=$(find "$OUT/pmc_$C" -name 'pmc_counter_collection.csv' | head -1)
  mkdir -p "$OUT/pmc_r1_$C"
  [ -n "$CSV" ] && cp "$CSV" "$OUT/pmc_r1_$C/pmc_counter_collection.csv"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" "$OUT/pmc_traffic.json"
# the bench line once more, now that a PMC pass at THESE kernel sources exists (bench.py quotes roofline.traffic only from a summary whose csrc_sha1 is the tree's):
# the summary goes where bench.py looks (profiles/, under the round's name) on this box; the caller copies it there in the repo as well
ROUND=$(echo "$TAG" | cut -c1-3)
cp "$OUT/pmc_traffic.json" "$ROOT/profiles/${ROUND}_pmc_traffic.json"
timeout 900 python "$ROOT/bench.py" > "$OUT/bench_with_traffic.json" 2> /dev/null
# keep the merge-back small: raw traces stay on the box
rm -rf "$OUT/prof" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES"
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
ls -la "$OUT"
