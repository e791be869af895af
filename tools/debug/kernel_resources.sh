#!/bin/bash
# Registers, spills, LDS and scratch of every kernel in the product library whose (mangled) name matches $1 (default: the
# tick launch's table kernel).  usage: tools/debug/kernel_resources.sh [pattern]
set -e
cd "$(dirname "$0")/../../beatrice-vst_amd/csrc"
pat=${1:-table_kernel}
tmp=$(mktemp -d)
cp libbeatrice_hip.so "$tmp/"
( cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading libbeatrice_hip.so > /dev/null 2>&1
  for f in libbeatrice_hip.so.*gfx950; do
    /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null |
      grep -E "^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.group_segment_fixed_size|\.vgpr_spill_count|\.sgpr_count" | paste - - - - - - |
      grep -E "$pat" | sed -E 's/(_Z[A-Za-z0-9_]{40})[A-Za-z0-9_]+/\1.../; s/ +/ /g'
  done )
rm -rf "$tmp"
