#!/bin/bash
# per-frame time of the cfg5 bench and its ray kernel (for build_experiment.sh)
python tools/dynamic_bench.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DYN', d['ms_per_frame'], d['kernels_ms_per_frame']['trace_rays_kernel'], d['mean_rgb'])"
