#!/bin/bash
# same-box A/B of two builds of the library at batch 1 (the reference's operating mode) and batch 64:
#   scripts/ab_lib_b1.sh <other .so>      (interleaved, three rounds)
OTHER=$1
one() { env $1 python bench.py --no-cpu-baseline --no-end-to-end --no-long $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['value'],1), 'pairs/s', round(d['ms_per_frame'],4), 'ms/frame', d['verified'])"; }
for i in 1 2 3; do
  one SN_X=0 "--batch 1 --steps 400 --warmup 20" "b1 this "
  one STEREONET_HIP_LIB=$OTHER "--batch 1 --steps 400 --warmup 20" "b1 other"
done
for i in 1 2; do
  one SN_X=0 "--steps 40" "b64 this "
  one STEREONET_HIP_LIB=$OTHER "--steps 40" "b64 other"
done
