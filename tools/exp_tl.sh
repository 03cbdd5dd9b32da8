mkdir -p gpurun_out/r3t; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -k "build or tree or refit or golden or cfg2 or cfg4 or animate or soak or scene" 2>&1 | tail -3
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r3t/tl -o tl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/build_timeline.py steps > $GRAFT_REPO_ROOT/gpurun_out/r3t/tl.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/build_timeline.py show gpurun_out/r3t/tl > gpurun_out/r3t/timeline.txt 2>&1
cat gpurun_out/r3t/timeline.txt
rm -rf gpurun_out/r3t/tl
for i in 1 2; do python bench.py --no-cpu-baseline --no-live-counters --no-dynamic --no-sort-bench | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['build_ms'], d['trace_ms'], d['build_Mtri_s'], d['value'], d['build_kernels_ms'])"; done
