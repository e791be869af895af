#!/bin/bash
# SQ counters of the tick launch (full ticks only): where do the wavefronts of a tick spend their cycles?
#   tools/debug/tick_sq_pmc.sh  -> gpurun_out/tick_sq_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/tick_sq_pmc.txt
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $ROOT/bench.py --steps 120 --warmup 10 --no-extras > /dev/null 2>&1
  python - "$(find /tmp/pmc -name '*counter_collection.csv' | head -1)" >> $OUT <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "table_kernel" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    v = sorted(v); v = v[len(v) // 2:]          # full ticks
    print("%-28s %14.0f per full tick (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
cat $OUT
