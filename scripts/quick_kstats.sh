#!/bin/bash
# Per-kernel times of the timed workload, streams serialised: scripts/quick_kstats.sh <tag> [ENV=val ...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env SN_NO_OVERLAP=1 "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_ser -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-long > /dev/null 2>&1
python $ROOT/scripts/kstats.py $(find $OUT/${TAG}_prof_ser -name "*kernel_stats.csv" | head -1) 30 > $OUT/${TAG}_kernel_summary_serialised.txt
rm -rf $OUT/${TAG}_prof_ser
cat $OUT/${TAG}_kernel_summary_serialised.txt
