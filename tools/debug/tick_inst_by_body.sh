#!/bin/bash
# Instructions per BODY TYPE of the tick launch: the launch with only one group of body types in its table
# (measurement build: `tools/debug/build_variant.sh meas -DBEATRICE_HIP_MEASUREMENT_BUILD`), SQ counters per dispatch.
#   tools/debug/tick_inst_by_body.sh [streams] [hops]  -> gpurun_out/tick_inst_by_body.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export BEATRICE_HIP_LIB=$ROOT/build_variants/lib_meas.so
B=${1:-256}; H=${2:-2}
OUT=$ROOT/gpurun_out/tick_inst_by_body.txt
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/debug/tick_inst_by_body.py $B $H > /tmp/groups_alone.txt 2>/tmp/groups_alone.err || { tail -5 /tmp/groups_alone.err; exit 1; }
: > /tmp/pmc_rows.txt
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $ROOT/tools/debug/tick_inst_by_body.py $B $H > /dev/null 2>&1
  python - "$(find /tmp/pmc -name '*counter_collection.csv' | head -1)" >> /tmp/pmc_rows.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)   # dispatch id -> counter -> value
for r in csv.DictReader(open(sys.argv[1])):
    if "table_kernel" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
N_FILL, N_MEAS = 30, 8
per = N_FILL + N_MEAS
ng = len(ids) // per
for g in range(ng):
    sl = ids[g * per + N_FILL:(g + 1) * per]
    acc = collections.defaultdict(float)
    for i in sl:
        for k, v in rows[i].items(): acc[k] += v / len(sl)
    print(g, " ".join("%s=%.0f" % kv for kv in sorted(acc.items())))
PY
done
python - /tmp/groups_alone.txt /tmp/pmc_rows.txt $B $H > $OUT <<'PY'
import sys, collections
names, alone = [], []
for l in open(sys.argv[1]):
    if l.startswith("group "):
        names.append(l[6:17].strip()); alone.append(float(l.split()[-5]))
c = collections.defaultdict(dict)
for l in open(sys.argv[2]):
    p = l.split()
    for kv in p[1:]:
        k, v = kv.split("="); c[int(p[0])][k] = float(v)
B, H = int(sys.argv[3]), int(sys.argv[4])
print("tick launch, %d streams x %d hops per step: SQ counters per FULL tick with only one group of body types in the table" % (B, H))
print("(measurement build; us alone = the launch's duration with only that group, HIP events, no profiler)")
hdr = ("group", "us alone", "waves", "MFMA", "VALU-MFMA", "per MFMA", "LDS", "VMEM", "SALU", "MFMA busy cyc", "LDS conflict", "LDS idx act", "wait inst any", "wave cycles")
print("%-10s %8s %7s %9s %10s %8s %9s %8s %9s %13s %12s %11s %13s %12s" % hdr)
tot = collections.defaultdict(float)
for g, n in enumerate(names):
    d = c.get(g, {})
    mf, va = d.get("SQ_INSTS_MFMA", 0), d.get("SQ_INSTS_VALU", 0)
    row = (n, alone[g], d.get("SQ_WAVES", 0), mf, va - mf, (va - mf) / mf if mf else float("nan"), d.get("SQ_INSTS_LDS", 0), d.get("SQ_INSTS_VMEM", 0), d.get("SQ_INSTS_SALU", 0),
           d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_LDS_IDX_ACTIVE", 0), d.get("SQ_WAIT_INST_ANY", 0), d.get("SQ_WAVE_CYCLES", 0))
    print("%-10s %8.2f %7.0f %9.0f %10.0f %8.2f %9.0f %8.0f %9.0f %13.0f %12.0f %11.0f %13.0f %12.0f" % row)
    if not n.startswith("all"):
        for i, v in enumerate(row[2:], 2):
            if i != 5: tot[i] += v
print("%-10s %8s %7.0f %9.0f %10.0f %8.2f %9.0f %8.0f %9.0f %13.0f %12.0f %11.0f %13.0f %12.0f" % ("sum groups", "", tot[2], tot[3], tot[4], tot[4] / tot[3], tot[6], tot[7], tot[8], tot[9], tot[10], tot[11], tot[12], tot[13]))
PY
cat $OUT
