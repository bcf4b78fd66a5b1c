cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in ${PROBE_VARIANTS:-SN_X=0 SN_F32_GRID=512 SN_F32_GRID=128}; do
  for d in ${PROBE_DILS:-1 8}; do
  rm -rf /tmp/pp
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $R/scripts/f32_tower_probe.py $d > /dev/null 2>&1
  echo "== $v dil $d"; python $R/scripts/kstats.py $(find /tmp/pp -name "*kernel_stats.csv" | head -1) 30 | grep -i "ref_conv_f32"
  done
done
