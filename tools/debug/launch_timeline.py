"""Timeline of the last N tick launches in a rocprofv3 kernel trace: start (us after the first of them), duration, idle gap before each; and what else ran
between the first and the last of them.   python tools/debug/launch_timeline.py <k_kernel_trace.csv> <N> [--brief] [--skip=K: the N launches after the first K instead]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
n = int(sys.argv[2])
ticks = [r for r in rows if "table_kernel" in r[2]]
skip = [int(a[7:]) for a in sys.argv if a.startswith("--skip=")]
ticks = ticks[skip[0]:skip[0] + n] if skip else ticks[-n:]
t0, t1 = ticks[0][0], ticks[-1][1]
print("%d launches, sum %.0f us, span %.0f us" % (len(ticks), sum(e - s for s, e, _ in ticks) / 1e3, (t1 - t0) / 1e3))
others = [r for r in rows if "table_kernel" not in r[2] and r[0] >= t0 - 3000000 and r[1] <= t1 + 300000]
for s, e, k in others:
    print("  other: start %8.1f dur %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, k[:70]))
if "--brief" not in sys.argv:
    prev = None
    for i, (s, e, _) in enumerate(ticks):
        print("  launch %2d  start %7.1f  dur %6.1f  gap %5.1f" % (i, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
        prev = e
