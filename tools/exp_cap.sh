mkdir -p gpurun_out/r3g
for c in 48 96 160 256; do echo "cap $c"; LBVH_LIB=build_exp/liblbvh_cap$c.so python tools/trace_only.py --no-check --reps 6 | tail -3; done > gpurun_out/r3g/cap.txt 2>&1
echo "no cap" >> gpurun_out/r3g/cap.txt; python tools/trace_only.py --no-check --reps 6 | tail -3 >> gpurun_out/r3g/cap.txt 2>&1
python tools/tile_costs.py >> gpurun_out/r3g/cap.txt 2>&1
cat gpurun_out/r3g/cap.txt
