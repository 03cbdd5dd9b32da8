mkdir -p gpurun_out/r3w
python -m pytest tests -m gpu -x -q > gpurun_out/r3w/pytest.txt 2>&1; tail -3 gpurun_out/r3w/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r3w/bench.json 2> gpurun_out/r3w/bench.err; tail -2 gpurun_out/r3w/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r3w/bench.json'))
print(d['value'], d['build_Mtri_s'], d['ms_per_step'], d['build_ms'], d['trace_ms'])
r=d['roofline']; print({k:r[k] for k in ('kernel_ms','bound','achieved','peak','frac')}); print({k:v for k,v in r['valu_issue'].items() if k!='basis'}); print(r['traffic'])
s=d['roofline_sort_scatter']; print({k:s[k] for k in ('achieved','frac','frac_of_measured_copy','kernel_ms','sort_Gkeys_s','traffic')})
print(d['cpu_baseline']); print(d['trace_variants_ms']); print(d['cfg5_dynamic']['ms_per_frame'], {k: v for k, v in d['cfg5_dynamic']['kernels_ms_per_frame'].items() if 'rays' in k or 'collapse' in k})
"
bash tools/prof.sh n > gpurun_out/r3w/prof.txt 2>&1; tail -25 gpurun_out/r3w/prof.txt
python tools/shard_times.py > gpurun_out/r3w/shard_times.txt 2>&1; cat gpurun_out/r3w/shard_times.txt
