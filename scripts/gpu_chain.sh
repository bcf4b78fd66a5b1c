#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "feature_blocks_chain or lowres_split_conv3x3" > gpurun_out/r05e_tests1.log 2>&1; echo "tests1 rc=$?"; tail -5 gpurun_out/r05e_tests1.log
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_f16.py -m gpu -x -q -k "FEAT_CHAIN or env13 or env14 or forward_small or full_size_epe or race_screen" > gpurun_out/r05e_tests2.log 2>&1; echo "tests2 rc=$?"; tail -5 gpurun_out/r05e_tests2.log
AB_STEPS=20 timeout 900 bash scripts/ab_env.sh SN_X=1 SN_FEAT_CHAIN=0 SN_X=2 SN_FEAT_CHAIN=0 SN_NO_OVERLAP=1 SN_NO_OVERLAP=1,SN_FEAT_CHAIN=0 --config=c5 --config=c5,SN_FEAT_CHAIN=0 > gpurun_out/r05_feat_chain_ab.txt 2>&1; cat gpurun_out/r05_feat_chain_ab.txt
timeout 400 bash scripts/quick_kstats.sh r05e > /dev/null 2>&1; grep -E "feat|total" gpurun_out/r05e_kernel_summary_serialised.txt
