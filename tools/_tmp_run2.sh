#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python3 -m pytest tests/test_timed_route_gpu.py -x -q -k "combiner" 2>&1 | grep -v "sun borders\|unrotation\|Y/X\|^file " | grep -B5 -A25 "Error\|assert" | head -80
