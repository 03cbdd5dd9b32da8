mkdir -p gpurun_out/r3q; export TMPDIR=/tmp; cd /tmp
LBVH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r3q/tl -o tl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/build_timeline.py steps > $GRAFT_REPO_ROOT/gpurun_out/r3q/tl.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/build_timeline.py show gpurun_out/r3q/tl > gpurun_out/r3q/timeline_nograph.txt 2>&1
cat gpurun_out/r3q/timeline_nograph.txt
rm -rf gpurun_out/r3q/tl
python bench.py --no-cpu-baseline --no-live-counters --no-dynamic --no-sort-bench | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph   ', d['ms_per_step'], d['build_ms'], d['trace_ms'])"
LBVH_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-live-counters --no-dynamic --no-sort-bench | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no graph', d['ms_per_step'], d['build_ms'], d['trace_ms'])"
