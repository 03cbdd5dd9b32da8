mkdir -p gpurun_out/r3r
python -m pytest tests/test_gpu_parity.py -x -q -k "trace or shard or cfg2 or golden or tie or equal_t or camera or soak or smoke" 2>&1 | tail -3
(echo "== product: ahead for classes >= 8"; python tools/trace_only.py --reps 5 | tail -6; python tools/shard_times.py; python tools/one_tile.py
for c in 0 6 99; do echo "== ahead class >= $c"; LBVH_LIB=build_exp/liblbvh_ahead$c.so python tools/trace_only.py --no-check --reps 5 | tail -3; LBVH_LIB=build_exp/liblbvh_ahead$c.so python tools/shard_times.py; done
echo "== never, one tile"; LBVH_LIB=build_exp/liblbvh_ahead99.so python tools/one_tile.py) > gpurun_out/r3r/ahead.txt 2>&1
cat gpurun_out/r3r/ahead.txt
