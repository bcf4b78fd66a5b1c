#!/bin/bash
# A/B helper: runs bench.py (short, no CPU baseline / end-to-end legs) once per argument; an argument is a
# comma-separated list of VAR=value settings and/or bench flags prefixed with "--", e.g.
#   bash scripts/ab_env.sh SN_FUSE=0 SN_FUSE=4,--refine-chunk=3
for spec in "$@"; do
  envs=""; flags=""
  IFS=',' read -ra parts <<< "$spec"
  for p in "${parts[@]}"; do
    if [[ $p == --* ]]; then flags="$flags ${p/=/ }"; else envs="$envs $p"; fi
  done
  echo "== $spec"
  env $envs python bench.py --steps ${AB_STEPS:-20} --no-cpu-baseline --no-end-to-end --no-long $flags 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], d['config']['refine_chunk'], d['config']['piece'])"
done
