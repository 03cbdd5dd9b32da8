# Round-end state on ONE box in ONE gpurun call: GPU tests, smoke, the driver's bench line, rocprofv3 summary + traffic of the same
# command, the one-GPU stand-in for the shares, static / cold / moving frames, a short fuzz soak, the sort's slow path.  Output: gpurun_out/r6w/ (copy what
# is to be judged into profiles/r6/ as n_*).
mkdir -p gpurun_out/r6w
python -m pytest tests -m gpu -x -q > gpurun_out/r6w/pytest.txt 2>&1; tail -3 gpurun_out/r6w/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r6w/bench.json 2> gpurun_out/r6w/bench.err; tail -2 gpurun_out/r6w/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r6w/bench.json'))
print(d['value'], d['build_Mtri_s'], d['ms_per_step'], d['build_ms'], d['trace_ms'])
r=d['roofline']; print({k:r[k] for k in ('kernel_ms','bound','achieved','peak','frac')}); print({k:v for k,v in r['valu_issue'].items() if k!='basis'}); print(r['traffic'])
s=d['roofline_sort_scatter']; print({k:s[k] for k in ('achieved','frac','frac_of_measured_copy','kernel_ms','sort_Gkeys_s','traffic')})
print(d['cpu_baseline']); print(d['trace_variants_ms']); c=d['cfg5_dynamic']; print(c['ms_per_frame'], {k: v for k, v in c['kernels_ms_per_frame'].items() if 'rays' in k or 'collapse' in k or 'animate' in k}, {k: c['roofline'][k] for k in ('kernel_ms','achieved','frac','traffic')})
"
python bench.py --gpus 2 --backend gloo --device 0 --steps 10 --no-sort-bench --no-cpu-baseline --no-dynamic --no-live-counters > gpurun_out/r6w/bench_two_ranks_one_gpu.json 2> gpurun_out/r6w/bench2.err; tail -1 gpurun_out/r6w/bench2.err
bash tools/prof.sh n6 > gpurun_out/r6w/prof.txt 2>&1; tail -25 gpurun_out/r6w/prof.txt
python tools/shard_times.py > gpurun_out/r6w/shard_times.txt 2>&1; cat gpurun_out/r6w/shard_times.txt
python tools/adapt_bench.py > gpurun_out/r6w/frames.txt 2>&1; cat gpurun_out/r6w/frames.txt
python tools/fuzz_parity.py 300 44 > gpurun_out/r6w/fuzz.txt 2>&1; tail -3 gpurun_out/r6w/fuzz.txt
# cfg4 (configs[3] on one GPU): the bench line and the rocprofv3 summary + FETCH / WRITE passes of the same command
python bench.py --workload cfg4 --no-dynamic --no-cpu-baseline --no-live-counters > gpurun_out/r6w/bench_cfg4.json 2> gpurun_out/r6w/bench_cfg4.err; tail -1 gpurun_out/r6w/bench_cfg4.err
bash tools/prof.sh n6cfg4 --workload cfg4 --no-dynamic --no-sort-bench > gpurun_out/r6w/prof_cfg4.txt 2>&1; tail -30 gpurun_out/r6w/prof_cfg4.txt

python tools/whole_frame_probe.py --cfg4 > gpurun_out/r6w/whole_frames.txt 2>&1; cat gpurun_out/r6w/whole_frames.txt
python tools/sort_cliff.py > gpurun_out/r6w/sort_cliff.txt 2>&1; cat gpurun_out/r6w/sort_cliff.txt
