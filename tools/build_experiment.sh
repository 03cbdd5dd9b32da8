#!/bin/bash
# tools/build_experiment.sh "<-D flags>" <file.hip> <python script + args>: rebuild one csrc file with experimental
# macros, run a script on the GPU box, restore the library.  Elimination experiments only (results may be wrong).
set -e
ROOT=/root/repo
CS=$ROOT/unitysimpleraytracing_amd/csrc
FLAGS="$1"; FILE="$2"; shift; shift
make -C $CS >/dev/null
cp $ROOT/unitysimpleraytracing_amd/liblbvh.so /tmp/liblbvh_good.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $FLAGS -I$ROOT/include -c $CS/$FILE.hip -o /tmp/exp_$FILE.o
OBJS=""
for f in lbvh_api lbvh_sort lbvh_build lbvh_trace lbvh_shade lbvh_path; do if [ $f = $FILE ]; then OBJS="$OBJS /tmp/exp_$FILE.o"; else OBJS="$OBJS $CS/$f.o"; fi; done
hipcc --offload-arch=gfx950 -shared -o $ROOT/unitysimpleraytracing_amd/liblbvh.so $OBJS
/usr/local/graft/bin/gpurun --timeout 600 -- "timeout 300 $*" 2>&1 | tail -12 || true
cp /tmp/liblbvh_good.so $ROOT/unitysimpleraytracing_amd/liblbvh.so
