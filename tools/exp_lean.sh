mkdir -p gpurun_out/r3j
for v in prio7 prio9; do echo $v; LBVH_LIB=build_exp/liblbvh_$v.so python tools/trace_only.py --no-check --reps 6 | tail -3; done > gpurun_out/r3j/trace.txt 2>&1
cat gpurun_out/r3j/trace.txt
