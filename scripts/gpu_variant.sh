#!/bin/bash
# A/B of a variant build of the library (scripts/build/libsn_$1.so) against the in-tree one: bit-equality of a 6-pair batch at
# 1280x720, bench, serialised kernel times of the named kernel
V=$1; MATCH=${2:-feat}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cat > /tmp/cmp_run.py <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from hobot_stereonet_amd import api, synth, weights
import tempfile, os
w, h, d, n = 1280, 720, 192, 6
xs = np.stack([synth.model_input_i8(w, h, d, 70 + i) for i in range(n)])
p = os.path.join(tempfile.mkdtemp(), "m.snw"); weights.save_snw(p, weights.synthetic(0), w, h, d)
with api.StereoNetHIP(p, max_batch=n, precision=api.PREC_F16) as eng:
    outs = [eng.infer(xs)[0] for _ in range(3)]
assert all(np.array_equal(outs[0], o) for o in outs[1:]), "not repeatable"
np.save(sys.argv[2], outs[0])
PY
python /tmp/cmp_run.py $ROOT /tmp/a.npy && STEREONET_HIP_LIB=$ROOT/scripts/build/libsn_$V.so python /tmp/cmp_run.py $ROOT /tmp/b.npy && python -c "
import numpy as np; a=np.load('/tmp/a.npy'); b=np.load('/tmp/b.npy'); print('variant == tree:', np.array_equal(a,b), 'max diff', float(np.abs(a-b).max()))"
for i in 1 2; do
  for lib in "" "STEREONET_HIP_LIB=$ROOT/scripts/build/libsn_$V.so"; do
    echo "== ${lib:-tree}"; env $lib python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --no-long 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], d['epe_vs_oracle_px'])"
  done
done
cd /tmp && export TMPDIR=/tmp
env SN_NO_OVERLAP=1 STEREONET_HIP_LIB=$ROOT/scripts/build/libsn_$V.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/var_prof -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-long > /dev/null 2>&1
python $ROOT/scripts/kstats.py $(find $OUT/var_prof -name "*kernel_stats.csv" | head -1) 30 | grep -E "$MATCH|total"
rm -rf $OUT/var_prof
