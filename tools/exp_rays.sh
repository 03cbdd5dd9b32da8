mkdir -p gpurun_out/r3s
python -m pytest tests/test_gpu_parity.py -x -q -k "ray or path or secondary or dynamic or animate or soak" 2>&1 | tail -4
for v in 1 2; do timeout 300 python tools/dynamic_bench.py | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_frame'], d['kernels_ms_per_frame'])"; done > gpurun_out/r3s/rays.txt 2>&1
cat gpurun_out/r3s/rays.txt
