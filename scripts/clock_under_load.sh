#!/bin/bash
# clock and package power while a bench runs: scripts/clock_under_load.sh <out> <bench args...>
OUT=$1; shift
python bench.py "$@" --no-cpu-baseline --no-end-to-end --no-verify --no-long > /dev/null 2>&1 &
LOADPID=$!
sleep ${LOAD_WARM_S:-25}
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power \(W\)"; sleep 1; done > $OUT 2>&1
wait $LOADPID
