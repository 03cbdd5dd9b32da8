#!/bin/bash
# Copies what tools/round_end.sh left under gpurun_out/ into profiles/r5/ as the judged round-end state (n_*).
set -e
P=profiles/r5
cp gpurun_out/r5w/bench.json $P/n_bench.json
cp gpurun_out/r5w/bench_two_ranks_one_gpu.json $P/n_bench_two_ranks_one_gpu.json
cp gpurun_out/r5w/shard_times.txt $P/n_shard_times_one_gpu.txt
cp gpurun_out/r5w/frames.txt $P/n_frames_static_cold_moving_shares.txt
(grep -v "^case [0-9]" gpurun_out/r5w/fuzz.txt; echo "(case lines dropped)") > $P/n_fuzz_soak.txt
cp gpurun_out/prof_n5/summary.txt $P/n_rocprofv3_summary.txt
cp gpurun_out/prof_n5/traffic.json $P/n_traffic.json
cp "$(find gpurun_out/prof_n5/stats -name '*kernel_stats.csv' | head -1)" $P/n_rocprofv3_kernel_stats.csv
tail -3 gpurun_out/r5w/pytest.txt
cp gpurun_out/r5w/bench_cfg4.json $P/n_bench_cfg4_one_gpu.json
cp gpurun_out/prof_n5cfg4/summary.txt $P/n_cfg4_rocprofv3_summary.txt
cp gpurun_out/prof_n5cfg4/traffic.json $P/n_cfg4_traffic.json
cp gpurun_out/r5w/bucket_sort_probe.txt $P/n_bucket_sort_probe.txt
cp gpurun_out/r5w/whole_frames.txt $P/n_whole_frames_cfg2_4k_cfg4.txt
