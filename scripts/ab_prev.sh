# A/B of the in-tree library against scripts/build/libsn_prev.so (a build of an earlier commit) on one box
B="python bench.py --no-cpu-baseline --no-end-to-end --steps 40 --warmup 3 $*"
run() { name=$1; shift; out=$("$@" 2>/dev/null | tail -1); python -c "
import json,sys
d=json.loads('''$out'''); print('$name', round(d['value'],1), d['verified'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms']*1e3,1))"; }
run new $B
run prev env STEREONET_HIP_LIB=$PWD/scripts/build/libsn_prev.so $B
run new $B
run prev env STEREONET_HIP_LIB=$PWD/scripts/build/libsn_prev.so $B
