#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/<prefix>_* (the files the design documents cite).
#   tools/install_profiles.sh r03d r03
set -e
TAG=${1:?tag}; PRE=${2:?prefix}
R=gpurun_out/$TAG
cp $R/bench.json profiles/${PRE}_bench.json
cp $R/kernel_stats_B256_tick.csv profiles/${PRE}_kernel_stats_B256_tick.csv
cp $R/kernel_stats_B256_in_order.csv profiles/${PRE}_kernel_stats_B256_in_order.csv
cp $R/tick_launch_durations.txt profiles/${PRE}_tick_launch_durations.txt
cp $R/pmc_traffic.json profiles/${PRE}_pmc_traffic.json
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do cp $R/pmc_r1_$C/per_kernel_mean.csv profiles/${PRE}_pmc_${C}_per_kernel_mean.csv; done
cp $R/b1/kernel_stats_B1.csv profiles/${PRE}_b1_kernel_stats.csv
cp $R/b1/b1_gaps.txt profiles/${PRE}_b1_per_kernel.txt
for f in bench_B1024 bench_B4096 bench_config3; do [ -f $R/$f.json ] && cp $R/$f.json profiles/${PRE}_$f.json; done
[ -f $R/bench_driver_flags.json ] && cp $R/bench_driver_flags.json profiles/${PRE}_bench_driver_flags_steps20.json
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
print("traffic quoted by bench.py now:", bench.pmc_traffic("tick", 256))
PY
