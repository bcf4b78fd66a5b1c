#!/bin/bash
# What do k_refin_f16's LDS bank conflicts cost?  (SQ_LDS_BANK_CONFLICT is 39 % of that kernel's LDS
# cycles — its staging stores are four 16-byte slots per thread at a 64-byte stride.)
# Throw-away build, NOT in the tree: a copy of csrc/ whose commit() stores slot k of thread t at (256 k + t) % BUF —
# consecutive lanes, consecutive slots: conflict-free, wrong results — timed against the shipping library with the
# streams serialised (rocprofv3 kernel stats of the same bench command).  Run on the GPU box from the repo root:
#   bash scripts/refin_conflict_probe.sh > gpurun_out/r05_refin_conflict_probe.txt
set -u
ROOT=$(pwd)
TMP=$(mktemp -d /tmp/sn_refin_probe.XXXX)
cp -r hobot_stereonet_amd/csrc $TMP/csrc
mkdir -p $TMP/include && cp include/stereonet_hip.h $TMP/include/
sed -i 's|      buf\[ur \* T::COLS + 4 \* uq + k\] = \*reinterpret_cast<const uint4\*>(&sl);|      buf[(256 * k + tid) % T::BUF] = *reinterpret_cast<const uint4*>(\&sl);   /* PROBE: conflict-free, wrong */|' $TMP/csrc/sn_kernels.hpp
grep -c "PROBE: conflict-free" $TMP/csrc/sn_kernels.hpp | sed 's/^/patched lines: /'
sed -i 's|#include "../../include/stereonet_hip.h"|#include "'$TMP'/include/stereonet_hip.h"|' $TMP/csrc/*.hip $TMP/csrc/sn_internal.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I $TMP/include -ldl -lpthread \
  -o $TMP/libsn_probe.so $TMP/csrc/stereonet_hip.hip $TMP/csrc/sn_mgpu.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for v in ship probe ship probe; do
  lib=""; [ $v = probe ] && lib="STEREONET_HIP_LIB=$TMP/libsn_probe.so"
  rm -rf $TMP/prof
  env SN_NO_OVERLAP=1 $lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $TMP/prof -- \
    python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-long --no-verify > /dev/null 2>&1
  echo "== $v"
  python $ROOT/scripts/kstats.py $(find $TMP/prof -name "*kernel_stats.csv" | head -1) 30 | grep -E "k_refin_f16|total kernel"
done
rm -rf $TMP
