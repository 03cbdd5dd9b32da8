#!/bin/bash
# Copies what tools/round_end.sh left under gpurun_out/ into profiles/r6/ as the judged round-end state (n_*).
set -e
P=profiles/r6
cp gpurun_out/r6w/bench.json $P/n_bench.json
cp gpurun_out/r6w/bench_two_ranks_one_gpu.json $P/n_bench_two_ranks_one_gpu.json
cp gpurun_out/r6w/shard_times.txt $P/n_shard_times_one_gpu.txt
cp gpurun_out/r6w/frames.txt $P/n_frames_static_cold_moving_shares.txt
(grep -v "^case [0-9]" gpurun_out/r6w/fuzz.txt; echo "(case lines dropped)") > $P/n_fuzz_soak.txt
cp gpurun_out/prof_n6/summary.txt $P/n_rocprofv3_summary.txt
cp gpurun_out/prof_n6/traffic.json $P/n_traffic.json
cp "$(find gpurun_out/prof_n6/stats -name '*kernel_stats.csv' | head -1)" $P/n_rocprofv3_kernel_stats.csv
tail -3 gpurun_out/r6w/pytest.txt
cp gpurun_out/r6w/bench_cfg4.json $P/n_bench_cfg4_one_gpu.json
cp gpurun_out/prof_n6cfg4/summary.txt $P/n_cfg4_rocprofv3_summary.txt
cp gpurun_out/prof_n6cfg4/traffic.json $P/n_cfg4_traffic.json
cp gpurun_out/r6w/sort_cliff.txt $P/n_sort_cliff.txt
cp gpurun_out/r6w/pytest.txt $P/n_pytest_gpu.txt
cp gpurun_out/r6w/whole_frames.txt $P/n_whole_frames_cfg2_4k_cfg4.txt
