cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in "SN_X=0" "SN_HEAD_NOSTAT=1" "SN_HEAD_TH=6" "SN_HEAD_TH=6 SN_HEAD_NOSTAT=1" "SN_HEAD_MFMA32=0" "SN_HEAD_MFMA32=0 SN_HEAD_NOSTAT=1"; do
  rm -rf /tmp/pp
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $R/bench.py --precision fp32 --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-long --no-verify > /dev/null 2>&1
  echo "== $v"; python $R/scripts/kstats.py $(find /tmp/pp -name "*kernel_stats.csv" | head -1) 30 | grep -i "head_final"
done
