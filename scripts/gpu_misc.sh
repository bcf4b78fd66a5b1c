#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_small or full_size or padded or conv2d or features_identical" > gpurun_out/r05f_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05f_tests.log
python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --batch 1 --steps 100 --warmup 5 > gpurun_out/r05_fp32_b1_bench.json 2>/dev/null
python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --steps 5 --batch 16 > gpurun_out/r05_fp32_b16_bench.json 2>/dev/null
for f in fp32_b1 fp32_b16; do python -c "
import json; d=json.loads(open('gpurun_out/r05_${f}_bench.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_frame'],3), d['verified'], round(d['roofline']['frac'],3))"; done
