#!/bin/bash
# round-6 same-box A/Bs: piece size, feature-layer path (5 interleaved runs), F16X3 workgroups per CU
OUT=gpurun_out
AB_STEPS=40 bash scripts/ab_env.sh SN_X=0 --piece=32 --piece=64 SN_X=0 --piece=32 --piece=64 > $OUT/r06_piece_sweep.txt 2>&1
AB_STEPS=40 bash scripts/ab_env.sh SN_X=0 SN_FEAT_DMA=0 SN_X=0 SN_FEAT_DMA=0 SN_X=0 SN_FEAT_DMA=0 SN_X=0 SN_FEAT_DMA=0 SN_X=0 SN_FEAT_DMA=0 > $OUT/r06_feat_ab.txt 2>&1
AB_STEPS=10 bash scripts/ab_env.sh --precision=f16x3 --precision=f16x3,SN_X3_WPC=2 --precision=f16x3 --precision=f16x3,SN_X3_WPC=2 > $OUT/r06_x3_wpc_ab.txt 2>&1
cat $OUT/r06_piece_sweep.txt $OUT/r06_feat_ab.txt $OUT/r06_x3_wpc_ab.txt
