#!/bin/bash
# A/B of compiler flags for the traversal kernel (both builds in one call, same box)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for EXTRA in "" "-fno-slp-vectorize" "-fno-slp-vectorize -mllvm -amdgpu-early-inline-all=true"; do
  make -C unitysimpleraytracing_amd/csrc clean >/dev/null; make -C unitysimpleraytracing_amd/csrc -j8 CXXFLAGS="$BASE $EXTRA" 2>&1 | grep -E "error"
  echo "flags [$EXTRA]: $(python tools/trace_only.py --reps 6 --no-check 2>&1 | grep 'trace ms' | awk '{print $3}' | sort -n | head -3 | tr '\n' ' ')"
done
