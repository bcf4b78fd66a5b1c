#!/bin/bash
# Regenerates the judged measurement artefacts on the GPU box (run through gpurun from the repo root):
#   scripts/refresh_profiles.sh <tag>        e.g. r02a
# -> gpurun_out/<tag>_*: bench line (default mode, with cpu_baseline + end_to_end), rocprofv3 kernel stats of the SAME
#    command (rocprofv3 serialises the dispatches, so these are per-kernel times, not the overlapped pipeline), separate
#    --pmc passes: FETCH_SIZE / WRITE_SIZE (scripts/pmc_traffic.py) and SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE
#    (scripts/mfma_busy.py), SQ_LDS_BANK_CONFLICT + SQ_LDS_IDX_ACTIVE (scripts/lds_conflicts.py); PAIRS_PER_LAUNCH = the --refine-chunk of the PMC passes: 4, two chunks per 8-pair batch, comparable across rounds), bench lines of the other precision modes / configs / the stream mode, the MALL probe.
# (the --pmc passes force --precision f16: an AUTO handle's first call also runs ONE pair in both arithmetics, whose one-pair
# launches would be averaged into the per-launch figures)
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
TAG=${1:-r02x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout 600"
$T python $ROOT/bench.py --cpu-baseline-torch > $OUT/${TAG}_f16_b64_bench.json 2> $OUT/${TAG}_bench.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-long > $OUT/${TAG}_bench_under_rocprof.log 2>&1
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_f16_b64_kernel_stats.csv
# the same with the streams serialised (SN_NO_OVERLAP=1): per-kernel durations as bench.py's profiling pass sees them
SN_NO_OVERLAP=1 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_ser -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-long > /dev/null 2>&1
python $ROOT/scripts/kstats.py $(find $OUT/${TAG}_prof_ser -name "*kernel_stats.csv" | head -1) 30 > $OUT/${TAG}_kernel_summary_serialised.txt
rm -rf $OUT/${TAG}_prof_ser
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- python $ROOT/bench.py --steps 2 --warmup 1 --batch 8 --refine-chunk 4 --precision f16 --no-cpu-baseline --no-end-to-end --no-verify --no-long > $OUT/${TAG}_pmc_$c.log 2>&1
done
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -- python $ROOT/bench.py --steps 2 --warmup 1 --batch 8 --refine-chunk 4 --precision f16 --no-cpu-baseline --no-end-to-end --no-verify --no-long > $OUT/${TAG}_pmc_mfma.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_lds -- python $ROOT/bench.py --steps 2 --warmup 1 --batch 8 --refine-chunk 4 --precision f16 --no-cpu-baseline --no-end-to-end --no-verify --no-long > $OUT/${TAG}_pmc_lds.log 2>&1
cd $ROOT
python scripts/lds_conflicts.py $(find $OUT/${TAG}_pmc_lds -name "*counter_collection.csv" | head -1) $OUT/${TAG}_lds_conflicts.json > $OUT/${TAG}_lds_conflicts.txt 2>&1
python scripts/pmc_traffic.py $(find $OUT/${TAG}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find $OUT/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic.json k_ref_block_stream_f16 ${PAIRS_PER_LAUNCH:-4} ", true>" > $OUT/${TAG}_pmc_summary.txt 2>&1
python scripts/mfma_busy.py $(find $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/${TAG}_mfma_busy.json > $OUT/${TAG}_mfma_busy.txt 2>&1
# round 5: same-box A/B of the low-resolution changes through the library's switches ("SN_X=0" = the default tree; the first
# switch line = the round-4 kernels: two down-conv kernels, per-layer k_conv_x3s feature launches, k_head_softargmin)
AB_STEPS=20 bash scripts/ab_env.sh SN_X=0 SN_DOWN01=0,SN_FEAT_DMA=0,SN_HEAD_FOLD=0 SN_DOWN01=0 SN_FEAT_DMA=0 SN_HEAD_FOLD=0 SN_X=0 SN_DOWN01=0,SN_FEAT_DMA=0,SN_HEAD_FOLD=0 > $OUT/${TAG}_lowres_ab.txt 2>&1
$T python bench.py --precision f16x3 --no-cpu-baseline --no-end-to-end --steps 20 > $OUT/${TAG}_f16x3_b64_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --steps 5 --batch 16 > $OUT/${TAG}_fp32_b16_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --no-cpu-baseline --no-end-to-end --batch 1 --steps 400 --warmup 20 > $OUT/${TAG}_f16_b1_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --config c5 --steps 50 > $OUT/${TAG}_c5_f16_b64_bench.json 2>> $OUT/${TAG}_bench.err          # hierarchical refinement (c5 default)
$T python bench.py --config c5 --refine single --steps 50 --no-cpu-baseline --no-end-to-end > $OUT/${TAG}_c5_single_f16_b64_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --refine multi --steps 40 --no-cpu-baseline --no-end-to-end > $OUT/${TAG}_c2_multi_f16_b64_bench.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --gpus 2 --dist-backend gloo --device-map 0,0 --batch 16 --steps 10 --no-cpu-baseline --no-end-to-end > $OUT/${TAG}_two_ranks_one_gpu_gloo.json 2>> $OUT/${TAG}_bench.err   # functional N > 1 run
$T python bench.py --stream 10 --batch 16 > $OUT/${TAG}_stream_c2_b16.json 2>> $OUT/${TAG}_bench.err
$T python bench.py --config c5 --stream 10 --batch 16 > $OUT/${TAG}_stream_c5_b16.json 2>> $OUT/${TAG}_bench.err
python scripts/node_bench.py 800 > $OUT/${TAG}_node_bench_publish1.json 2>> $OUT/${TAG}_bench.err
python scripts/node_bench.py 800 STEREONET_PUB_OUTPUT=0 > $OUT/${TAG}_node_bench_publish0.json 2>> $OUT/${TAG}_bench.err
[ -x scripts/build/mall_probe ] && ./scripts/build/mall_probe > $OUT/${TAG}_mall_probe.txt 2>&1
# clock and package power while the timed workload runs (the tower runs at the 1400 W cap)
python bench.py --steps 2000 --no-cpu-baseline --no-end-to-end --no-verify > /dev/null 2>&1 &
LOADPID=$!
sleep 25
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power \(W\)"; sleep 1; done > $OUT/${TAG}_clock_power_under_load.txt 2>&1
wait $LOADPID
[ -x scripts/build/stream_block_probe ] && ./scripts/build/stream_block_probe 300 > $OUT/${TAG}_stream_block_probe.txt 2>&1
python scripts/kstats.py $OUT/${TAG}_f16_b64_kernel_stats.csv 30 > $OUT/${TAG}_kernel_summary.txt
python scripts/tower_sequence.py $(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_tower_sequence.txt 2>&1
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_lds     # raw traces: scratch
cat $OUT/${TAG}_mfma_busy.txt | head -12
$T python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --batch 1 --steps 100 --warmup 5 > $OUT/${TAG}_fp32_b1_bench.json 2>> $OUT/${TAG}_bench.err
for f in f16_b64 f16x3_b64 fp32_b16 fp32_b1 f16_b1 c5_f16_b64 c5_single_f16_b64 c2_multi_f16_b64; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_${f}_bench.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"],1), d["unit"], "ms/frame", round(d["ms_per_frame"],3), "verified", d["verified"], "roofline", d["roofline"]["bound"], round(d["roofline"]["frac"],3), "e2e", (d.get("end_to_end") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f ERR", e)
PY
done
for f in stream_c2_b16 stream_c5_b16 two_ranks_one_gpu_gloo; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_${f}.json").read().strip().splitlines()[-1]); print("$f", round(d["value"],1), d["unit"], d["timed_seconds"])
except Exception as e:
    print("$f ERR", e)
PY
done
tail -5 $OUT/${TAG}_bench.err
