#!/bin/bash
# tools/sort_experiment.sh "<-D flags>" [sizes...]: build liblbvh.so with experimental macros in lbvh_sort.hip, run
# tools/sort_bench.py on the GPU box without the sortedness check, restore the library.  Elimination experiments only.
set -e
ROOT=/root/repo
CS=$ROOT/unitysimpleraytracing_amd/csrc
FLAGS="$1"; shift
make -C $CS >/dev/null
cp $ROOT/unitysimpleraytracing_amd/liblbvh.so /tmp/liblbvh_good.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $FLAGS -I$ROOT/include -c $CS/lbvh_sort.hip -o /tmp/sort_exp.o
hipcc --offload-arch=gfx950 -shared -o $ROOT/unitysimpleraytracing_amd/liblbvh.so $CS/lbvh_api.o /tmp/sort_exp.o $CS/lbvh_build.o $CS/lbvh_trace.o $CS/lbvh_shade.o $CS/lbvh_path.o
sed 's/    assert (k\[1:\] >= k\[:-1\]).all()/    pass/' $ROOT/tools/sort_bench.py > $ROOT/tools/_sort_bench_nocheck.py
/usr/local/graft/bin/gpurun --timeout 600 -- "timeout 300 python tools/_sort_bench_nocheck.py $*" 2>&1 | grep "2^\|tiles " || true
rm -f $ROOT/tools/_sort_bench_nocheck.py
cp /tmp/liblbvh_good.so $ROOT/unitysimpleraytracing_amd/liblbvh.so
