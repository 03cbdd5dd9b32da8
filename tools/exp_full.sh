mkdir -p gpurun_out/r3n
python -m pytest tests -m gpu -x -q -s -k "rocprim or ties or sharded or obj or prefilled" > gpurun_out/r3n/pytest_new.txt 2>&1; tail -12 gpurun_out/r3n/pytest_new.txt
bash tools/prof.sh cfg4 --workload cfg4 --no-dynamic --no-sort-bench > gpurun_out/r3n/prof_cfg4.txt 2>&1
tail -40 gpurun_out/r3n/prof_cfg4.txt
python bench.py --workload cfg4 --no-cpu-baseline --no-live-counters --no-dynamic --no-sort-bench > gpurun_out/r3n/bench_cfg4.json 2> gpurun_out/r3n/bench_cfg4.err; tail -c 1500 gpurun_out/r3n/bench_cfg4.json
