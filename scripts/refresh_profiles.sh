#!/bin/bash
# Regenerates the judged measurement artefacts on the GPU box (run through gpurun from the repo root):
#   scripts/refresh_profiles.sh <tag>        e.g. r01d
# -> gpurun_out/<tag>_*: bench line (default mode, with cpu_baseline), rocprofv3 kernel stats of the SAME command
#    (two-stream overlap on) and of a serialised run, separate --pmc FETCH_SIZE / WRITE_SIZE passes summarised by
#    scripts/pmc_traffic.py, bench lines of the other precision modes, PCIe-inclusive rates.
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
TAG=${1:-r01x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout 300"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $ROOT/bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_under_rocprof.log 2>&1
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_f16_b64_kernel_stats.csv
SN_NO_OVERLAP=1 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_ser -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_ser.log 2>&1
cp $(find $OUT/${TAG}_prof_ser -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_f16_b64_serialised_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  SN_NO_OVERLAP=1 $T rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- python $ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $ROOT
python scripts/pmc_traffic.py $(find $OUT/${TAG}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find $OUT/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic.json k_ref_conv_f16 2 > $OUT/${TAG}_pmc_summary.txt 2>&1
$T python bench.py > $OUT/${TAG}_f16_b64_bench.json 2> $OUT/${TAG}_bench.err
$T python bench.py --precision f16x3 --no-cpu-baseline --steps 5 > $OUT/${TAG}_f16x3_b64_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --precision fp32 --no-cpu-baseline --steps 3 --batch 16 > $OUT/${TAG}_fp32_b16_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --no-cpu-baseline --batch 1 --steps 200 --warmup 20 > $OUT/${TAG}_f16_b1_bench.json 2>> $OUT/${TAG}_bench.err
$T python scripts/pcie_rate.py > $OUT/${TAG}_pcie_rate.txt 2>&1
python scripts/kstats.py $OUT/${TAG}_f16_b64_serialised_kernel_stats.csv 30 > $OUT/${TAG}_serialised_summary.txt
tail -n 3 $OUT/${TAG}_pcie_rate.txt
for f in f16_b64 f16x3_b64 fp32_b16 f16_b1; do python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_${f}_bench.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],1), d["unit"], "ms/frame", round(d["ms_per_frame"],3), "roofline", d["roofline"]["bound"], round(d["roofline"]["frac"],3), "traffic", d["roofline"]["traffic"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
