#!/bin/bash
# per-rank share times at 1080p (for build_experiment.sh)
python tools/shard_times.py 2>&1 | tail -4
