#!/bin/bash
# fp32 (BASELINE configs[1]) measurement set: scripts/fp32_round.sh <tag>  -> gpurun_out/<tag>_fp32_*
set -u
TAG=${1:-r06x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
T="timeout 600"
$T python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --batch 1 --steps 100 --warmup 5 > $OUT/${TAG}_fp32_b1_bench.json 2> $OUT/${TAG}_fp32.err
$T python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --steps 5 --batch 16 > $OUT/${TAG}_fp32_b16_bench.json 2>> $OUT/${TAG}_fp32.err
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof32 -- python $ROOT/bench.py --precision fp32 --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --no-long > /dev/null 2>&1
python $ROOT/scripts/kstats.py $(find $OUT/${TAG}_prof32 -name "*kernel_stats.csv" | head -1) 30 > $OUT/${TAG}_fp32_b1_kernel_summary.txt
rm -rf $OUT/${TAG}_prof32
cd $ROOT
for f in fp32_b1 fp32_b16; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_${f}_bench.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"],1), d["unit"], "ms/frame", round(d["ms_per_frame"],3), "verified", d["verified"], "roofline", d["roofline"]["bound"], round(d["roofline"]["frac"],3))
except Exception as e:
    print("$f ERR", e)
PY
done
tail -3 $OUT/${TAG}_fp32.err
