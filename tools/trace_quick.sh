#!/bin/bash
# standalone frame times + parity line (for build_experiment.sh)
python tools/trace_only.py --reps 5 | tail -5
