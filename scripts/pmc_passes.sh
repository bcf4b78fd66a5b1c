#!/bin/bash
# Several rocprofv3 --pmc passes (kernel-trace only, a few SQ counters each) over a short bench run, summarised per kernel:
#   bash scripts/pmc_passes.sh <tag> "<group 1 counters>" "<group 2 counters>" ...
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0; files=""
for grp in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${TAG}_$i -- python $ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-end-to-end --no-verify --no-long > $OUT/${TAG}_pass$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then files="$files $f"; else echo "pass $i ($grp): no counters collected"; tail -3 $OUT/${TAG}_pass$i.log; fi
done
cd $ROOT
python scripts/pmc_generic.py $OUT/${TAG}_counters.json $files > $OUT/${TAG}_counters.txt
head -60 $OUT/${TAG}_counters.txt
