#!/bin/bash
# Copies what tools/round_end.sh left under gpurun_out/ into profiles/r4/ as the judged round-end state (n_*).
set -e
P=profiles/r4
cp gpurun_out/r4w/bench.json $P/n_bench.json
cp gpurun_out/r4w/bench_two_ranks_one_gpu.json $P/n_bench_two_ranks_one_gpu.json
cp gpurun_out/r4w/shard_times.txt $P/n_shard_times_one_gpu.txt
cp gpurun_out/r4w/frames.txt $P/n_frames_static_cold_moving_shares.txt
(grep -v "^case [0-9]" gpurun_out/r4w/fuzz.txt; echo "(case lines dropped)") > $P/n_fuzz_soak.txt
cp gpurun_out/prof_n/summary.txt $P/n_rocprofv3_summary.txt
cp gpurun_out/prof_n/traffic.json $P/n_traffic.json
cp "$(find gpurun_out/prof_n/stats -name '*kernel_stats.csv' | head -1)" $P/n_rocprofv3_kernel_stats.csv
tail -3 gpurun_out/r4w/pytest.txt
