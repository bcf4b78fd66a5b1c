set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; T="timeout 600"
python -m pytest tests -m gpu -x -q > $OUT/r06u_gputests.log 2>&1; grep -n "passed\|failed" $OUT/r06u_gputests.log | tail -2
python bench.py --precision f16x3 --no-cpu-baseline --no-end-to-end --steps 20 > $OUT/r06_f16x3_b64_bench.json 2> $OUT/r06u.err
python bench.py --config c5 --precision f16x3 --no-cpu-baseline --no-end-to-end --steps 20 > $OUT/r06_c5_f16x3_b64_bench.json 2>> $OUT/r06u.err
ARGS3="--steps 2 --warmup 1 --batch 8 --refine-chunk 4 --precision f16x3 --no-cpu-baseline --no-end-to-end --no-verify --no-long"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/r06_x3_pmc_$c -- python $ROOT/bench.py $ARGS3 > /dev/null 2>&1
done
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/r06_x3_pmc_mfma -- python $ROOT/bench.py $ARGS3 > /dev/null 2>&1
cd $ROOT
python scripts/pmc_traffic.py $(find $OUT/r06_x3_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/r06_x3_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/r06_x3_pmc_traffic.json k_ref_block_stream_x3 4 "@@none@@" > $OUT/r06_x3_pmc_summary.txt 2>&1
python scripts/mfma_busy.py $(find $OUT/r06_x3_pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/r06_x3_mfma_busy.json > $OUT/r06_x3_mfma_busy.txt 2>&1
rm -rf $OUT/r06_x3_pmc_FETCH_SIZE $OUT/r06_x3_pmc_WRITE_SIZE $OUT/r06_x3_pmc_mfma
python -c "
import json
for f in ('f16x3_b64','c5_f16x3_b64'):
    d=json.loads(open('$OUT/r06_%s_bench.json'%f).read().strip().splitlines()[-1]); r=d['roofline']; print(f, round(d['value'],1), d['verified'], round(r['frac'],3), r.get('executed_mfma_frac'), r['traffic'], r['avg_launch_ms'])
d=json.load(open('$OUT/r06_x3_pmc_traffic.json')); print(d['dominant_avg_hbm_bytes_per_launch'])"
head -5 $OUT/r06_x3_mfma_busy.txt
