set -u
TAG=r06; ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
T="timeout 600"
ARGS="--steps 2 --warmup 1 --batch 8 --refine-chunk 4 --precision f16 --no-cpu-baseline --no-end-to-end --no-verify --no-long"
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_$c.log 2>&1
done
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_mfma.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_lds -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_lds.log 2>&1
cd $ROOT
python scripts/lds_conflicts.py $(find $OUT/${TAG}_pmc_lds -name "*counter_collection.csv" | head -1) $OUT/${TAG}_lds_conflicts.json > $OUT/${TAG}_lds_conflicts.txt 2>&1
python scripts/pmc_traffic.py $(find $OUT/${TAG}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic.json k_ref_block_stream_f16 4 ", true>" > $OUT/${TAG}_pmc_summary.txt 2>&1
python scripts/mfma_busy.py $(find $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/${TAG}_mfma_busy.json > $OUT/${TAG}_mfma_busy.txt 2>&1
# the split mode's streamed block: fabric bytes and matrix-pipe occupancy
ARGS3="--steps 2 --warmup 1 --batch 8 --refine-chunk 4 --precision f16x3 --no-cpu-baseline --no-end-to-end --no-verify --no-long"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_x3_pmc_$c -- python $ROOT/bench.py $ARGS3 > /dev/null 2>&1
done
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_x3_pmc_mfma -- python $ROOT/bench.py $ARGS3 > /dev/null 2>&1
cd $ROOT
python scripts/pmc_traffic.py $(find $OUT/${TAG}_x3_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/${TAG}_x3_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_x3_pmc_traffic.json k_ref_block_stream_x3 4 "@@none@@" > $OUT/${TAG}_x3_pmc_summary.txt 2>&1
python scripts/mfma_busy.py $(find $OUT/${TAG}_x3_pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/${TAG}_x3_mfma_busy.json > $OUT/${TAG}_x3_mfma_busy.txt 2>&1
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_lds $OUT/${TAG}_x3_pmc_FETCH_SIZE $OUT/${TAG}_x3_pmc_WRITE_SIZE $OUT/${TAG}_x3_pmc_mfma
python -c "import json; d=json.load(open('$OUT/${TAG}_pmc_traffic.json')); print('f16', d['dominant_avg_hbm_bytes_per_launch'])"
python -c "import json; d=json.load(open('$OUT/${TAG}_x3_pmc_traffic.json')); print('x3', d['dominant_avg_hbm_bytes_per_launch'])"
head -8 $OUT/${TAG}_x3_mfma_busy.txt; head -6 $OUT/${TAG}_mfma_busy.txt
