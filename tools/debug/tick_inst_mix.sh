#!/bin/bash
# instruction mix of the tick launch per group of bodies: the launch with one group left out (BEATRICE_HIP_TICK_DROP),
# differences = that group's instructions per full tick.  -> gpurun_out/tick_inst_mix.txt
# (BEATRICE_HIP_TICK_DROP exists in measurement builds only: first `tools/debug/build_variant.sh meas -DBEATRICE_HIP_MEASUREMENT_BUILD`;
#  this script then runs on build_variants/lib_meas.so)
[ -f "$(dirname "$0")/../../build_variants/lib_meas.so" ] && export BEATRICE_HIP_LIB="$(cd "$(dirname "$0")/../.." && pwd)/build_variants/lib_meas.so"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/tick_inst_mix.txt
: > $OUT
for drop in ${DROPS:-0 1 2 4 8 16 32 64 128}; do
  rm -rf /tmp/pmc
  BEATRICE_HIP_TICK_DROP=$drop rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $ROOT/bench.py --steps 100 --warmup 10 --no-extras > /dev/null 2>&1
  python - "$(find /tmp/pmc -name '*counter_collection.csv' | head -1)" $drop >> $OUT <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "table_kernel" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
row = {}
for k, v in d.items():
    v = sorted(v); v = v[len(v) // 2:]
    row[k] = sum(v) / len(v)
print("drop %3s " % sys.argv[2] + " ".join("%s %.0f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), row[k]) for k in sorted(row)))
PY
done
cat $OUT
