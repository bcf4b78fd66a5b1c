#!/usr/bin/env python3
"""Summarises `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of a build) per kernel:
    hipcc --offload-arch=gfx950 -O3 ... -Rpass-analysis=kernel-resource-usage -o /tmp/x.so stereonet_hip.hip 2> ru.txt
    python scripts/resource_usage.py ru.txt
Columns: VGPRs, AGPRs, scratch bytes per lane (non-zero = spills), waves per SIMD, static LDS bytes, kernel."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
for b in blocks:
    name = b.split("\n")[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass

    def g(key):
        m = re.search(re.escape(key) + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    print("%4d v %4d a %5d scr %2d occ %6d lds  %s" % (g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"),
                                                     g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]"), name[:150]))
