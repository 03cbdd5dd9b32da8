#!/bin/bash
# tools/build_only_grep.sh <pattern>: per-kernel rebuild times filtered by a pattern (for build_experiment.sh)
python tools/build_only.py | grep -E "$1|build "
