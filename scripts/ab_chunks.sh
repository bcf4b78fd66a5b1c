B="python bench.py --no-cpu-baseline --no-end-to-end --steps 30 --warmup 3"
run() { name=$1; shift; out=$("$@" 2>/dev/null | tail -1); python -c "
import json,sys
d=json.loads('''$out'''); print('$name', round(d['value'],1), d['verified'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms']*1e3,1))"; }
run c2_rc1_ns2 $B
run c2_rc2_ns1 env SN_TOWER_STREAMS=1 $B --refine-chunk 2
run c2_rc1_ns2_again $B
run c2_rc2_ns1_again env SN_TOWER_STREAMS=1 $B --refine-chunk 2
run c5_rc1_ns2 $B --config c5
run c5_rc2_ns1 env SN_TOWER_STREAMS=1 $B --config c5 --refine-chunk 2
run c5_rc2_ns2 $B --config c5 --refine-chunk 2
run c5_rc4_ns1 env SN_TOWER_STREAMS=1 $B --config c5 --refine-chunk 4
run c5_rc4_ns2 $B --config c5 --refine-chunk 4
run c5_rc3_ns2 $B --config c5 --refine-chunk 3
