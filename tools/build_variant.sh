#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: compiles the WORKING TREE's csrc into build_exp/liblbvh_<name>.so
# (git-ignored, travels to the GPU box).  A/B measurements load a variant with LBVH_LIB=build_exp/liblbvh_<name>.so
# (unitysimpleraytracing_amd/_native.py); the product library unitysimpleraytracing_amd/liblbvh.so is never touched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
CS=$R/unitysimpleraytracing_amd/csrc
OUT=$R/build_exp
mkdir -p $OUT/obj_$NAME
for f in lbvh_api lbvh_sort lbvh_build lbvh_trace lbvh_shade lbvh_path; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" \
      -c $CS/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblbvh_$NAME.so $OUT/obj_$NAME/*.o
rm -rf $OUT/obj_$NAME
echo built $OUT/liblbvh_$NAME.so
