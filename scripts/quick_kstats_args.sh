#!/bin/bash
# Per-kernel times for arbitrary bench.py arguments: scripts/quick_kstats_args.sh <tag> <bench args...>
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-long "$@" > /dev/null 2>&1
python $ROOT/scripts/kstats.py $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) 30 > $OUT/${TAG}_kernel_summary.txt
rm -rf $OUT/${TAG}_prof
cat $OUT/${TAG}_kernel_summary.txt
