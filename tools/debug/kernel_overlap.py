"""tools/debug/kernel_overlap.py <rocprofv3 kernel_trace.csv> <kernel name part A> <kernel name part B>: how much of kernel A's run time lies inside runs of kernel B
(do the wrapper launches run beside the tick launches, or between them?)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
A = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if sys.argv[2] in r["Kernel_Name"]]
Bk = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if sys.argv[3] in r["Kernel_Name"])
tot = ov = 0
for s, e in A:
    tot += e - s
    for bs, be in Bk:
        if be <= s: continue
        if bs >= e: break
        ov += min(e, be) - max(s, bs)
print("%s: %d launches, mean %.1f us, %.1f %% of their time inside a %s launch" % (sys.argv[2], len(A), tot / max(len(A), 1) / 1e3, 100.0 * ov / max(tot, 1), sys.argv[3]))
if Bk:
    durs = [(e - s) / 1e3 for s, e in Bk]
    gaps = [(Bk[i + 1][0] - Bk[i][1]) / 1e3 for i in range(len(Bk) - 1)]
    gaps = [g for g in gaps if g < 2000]
    print("%s: %d launches, mean %.1f us, mean gap to the next %.1f us" % (sys.argv[3], len(Bk), sum(durs) / len(durs), sum(gaps) / max(len(gaps), 1)))
