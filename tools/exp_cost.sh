mkdir -p gpurun_out/r3u
python -m pytest tests/test_gpu_parity.py -x -q -k "tile_costs_exported or shard" 2>&1 | tail -5
python tools/moving_bench.py 8 exchange > gpurun_out/r3u/moving_exchange_8.txt 2>&1; cat gpurun_out/r3u/moving_exchange_8.txt
python tools/moving_bench.py 4 exchange > gpurun_out/r3u/moving_exchange_4.txt 2>&1; cat gpurun_out/r3u/moving_exchange_4.txt
