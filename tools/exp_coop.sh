mkdir -p gpurun_out/r3l
for rep in 1 2; do
echo "lean coop"; python tools/shard_times.py
echo "old coop"; LBVH_LIB=build_exp/liblbvh_oldcoop.so python tools/shard_times.py
done > gpurun_out/r3l/coop.txt 2>&1
cat gpurun_out/r3l/coop.txt
