#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_f16.py tests/test_gpu_parity.py -m gpu -x -q -k "down01 or lowres_split or forward_small or full_size_epe" > gpurun_out/r05c_tests1.log 2>&1; echo "tests1 rc=$?"; tail -4 gpurun_out/r05c_tests1.log
timeout 500 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "env13 or env14 or env15 or env16" > gpurun_out/r05c_tests2.log 2>&1; echo "tests2 rc=$?"; tail -4 gpurun_out/r05c_tests2.log
AB_STEPS=20 timeout 900 bash scripts/ab_env.sh SN_X=1 SN_HEAD_FOLD=0 SN_X=2 SN_HEAD_FOLD=0 > gpurun_out/r05c_ab.txt 2>&1; cat gpurun_out/r05c_ab.txt
timeout 400 bash scripts/quick_kstats.sh r05c > /dev/null 2>&1; head -22 gpurun_out/r05c_kernel_summary_serialised.txt
timeout 600 bash scripts/refin_conflict_probe.sh > gpurun_out/r05_refin_conflict_probe.txt 2>&1; cat gpurun_out/r05_refin_conflict_probe.txt
