#!/usr/bin/env python3
"""Instruction mix between the first and the last MFMA of one kernel in a hipcc -S device listing.
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o sn.s stereonet_hip.hip
    python scripts/asm_mix.py sn.s <mangled-name-substring>"""
import sys
from collections import Counter

L = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(L) if l.startswith("_Z") and sys.argv[2] in l and l.rstrip().split(":")[0].endswith(sys.argv[2].split()[-1] if False else "") and ":" in l]
for st in starts:
    end = next(i for i in range(st, len(L)) if "s_endpgm" in L[i])
    body = L[st:end + 1]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    if not mf:
        continue
    span = body[mf[0]:mf[-1] + 1]
    c = Counter()
    for l in span:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    print(L[st].split(":")[0][:120])
    print(f"  {len(mf)} MFMAs, {tot} instructions in the MFMA span = {(tot - len(mf)) / len(mf):.2f} other per MFMA")
    print("  " + ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
