#!/bin/bash
# tools/build_head_variant.sh: compiles csrc of the last COMMIT into build_exp/liblbvh_head.so, for same-box A/B runs of
# the working tree against it (LBVH_LIB=build_exp/liblbvh_head.so).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
WT=$(mktemp -d /tmp/lbvh_head.XXXX)
git -C $R worktree add -f $WT HEAD -q
mkdir -p $R/build_exp/obj_head
for f in lbvh_api lbvh_sort lbvh_build lbvh_trace lbvh_shade lbvh_path; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -c $WT/unitysimpleraytracing_amd/csrc/$f.hip -o $R/build_exp/obj_head/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_exp/liblbvh_head.so $R/build_exp/obj_head/*.o
rm -rf $R/build_exp/obj_head
git -C $R worktree remove --force $WT
echo built build_exp/liblbvh_head.so from $(git -C $R rev-parse --short HEAD)
