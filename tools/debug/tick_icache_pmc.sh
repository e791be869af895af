cd /tmp && export TMPDIR=/tmp
for set in "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
rm -rf /tmp/pm; rocprofv3 --pmc $set --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --steps 100 --warmup 5 > /dev/null 2>&1
python - "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "table_kernel" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    v=sorted(v); v=v[len(v)//2:]; print(k, sum(v)/len(v))
PY
done
