python bench.py --no-dynamic 2>&1 | tail -1 > gpurun_out/bench_n.json; python -c "
import json
d=json.load(open('gpurun_out/bench_n.json'))
print('RESULT', d['value'], d['build_Mtri_s'], d['ms_per_step'], d['build_ms'], d['trace_ms'], d['config']['hit_fraction'])
"
