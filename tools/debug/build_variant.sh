#!/bin/bash
# tools/debug/build_variant.sh <name> <extra hipcc flags...>  ->  build_variants/lib_<name>.so (another build of the product library)
set -e
name=$1; shift
cd "$(dirname "$0")/../../beatrice-vst_amd"
mkdir -p ../build_variants/obj_$name
for f in csrc/*.hip; do
  o=../build_variants/obj_$name/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I../include -Icsrc "$@" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build_variants/lib_$name.so ../build_variants/obj_$name/*.o
rm -rf ../build_variants/obj_$name
ls -la ../build_variants/lib_$name.so
