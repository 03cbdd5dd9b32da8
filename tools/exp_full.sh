mkdir -p gpurun_out/r3m
python -m pytest tests -m gpu -x -q > gpurun_out/r3m/pytest.txt 2>&1; tail -3 gpurun_out/r3m/pytest.txt
python bench.py > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err; tail -2 gpurun_out/r3m/bench.err
python tools/one_tile.py > gpurun_out/r3m/one_tile.txt 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r3m/bench.json'))
print(d['value'], d['build_Mtri_s'], d['ms_per_step'], d['build_ms'], d['trace_ms'])
r=d['roofline']; print({k:r[k] for k in ('kernel_ms','bound','achieved','peak','frac')}); print(r['valu_issue']); print(r['traffic'])
print(d['trace_variants_ms']); print(d['cfg5_dynamic']['ms_per_frame'])
"
