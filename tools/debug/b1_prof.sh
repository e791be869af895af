#!/bin/bash
# rocprofv3 kernel trace of the 1-stream C-ABI (plain launches, the default since round 5; BEATRICE_HIP_HOP_GRAPH=1 = the per-call graphs, which rocprofv3 does not survive).
# Output: gpurun_out/$1/{kernel_stats_B1.csv, b1_gaps.txt}
out=gpurun_out/${1:-b1}
mkdir -p $out
export TMPDIR=/tmp
BEATRICE_HIP_HOP_GRAPH=1 timeout 200 python tools/debug/b1_hops.py 2000 > $out/b1_graph.txt 2>&1
timeout 200 python tools/debug/b1_hops.py 2000 > $out/b1_eager.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b1 --output-format csv -- python tools/debug/b1_hops.py 400 > $out/b1_under_rocprof.txt 2>&1
cp $(find $out/prof -name '*kernel_stats.csv' | head -1) $out/kernel_stats_B1.csv
python - "$(find $out/prof -name '*kernel_trace.csv' | head -1)" > $out/b1_gaps.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
rows = rows[len(rows) // 2:]
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 20000]
print("kernels %d  busy %.1f us  mean kernel %.2f us  mean gap(<20us) %.2f us  n gaps >=20us %d" % (len(rows), busy / 1e3, busy / 1e3 / len(rows), sum(small) / 1e3 / max(1, len(small)), len(gaps) - len(small)))
per = collections.defaultdict(list)
for s, e, n in rows: per[n[:100]].append(e - s)
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])): print("%7d x %7.2f us  %s" % (len(v), sum(v) / len(v) / 1e3, n))
PY
rm -rf $out/prof
cat $out/b1_graph.txt $out/b1_eager.txt $out/b1_under_rocprof.txt | grep p50
head -5 $out/b1_gaps.txt
