#!/usr/bin/env python3
"""Node-level throughput: runs compat/build/node_harness --bench on a seeded 1280x720 stereo frame.
    python scripts/node_bench.py [frames] [ENV=val ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hobot_stereonet_amd import synth, weights  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
env = dict(os.environ, SN_LOG_LEVEL="3")
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        env[k] = v
w, h, d = 1280, 720, 192
td = tempfile.mkdtemp()
weights.save_snw(td + "/m.snw", weights.synthetic(0, 1), w, h, d)
synth.sbs_nv12_frame(w, h, d, 21).tofile(td + "/s.bin")
r = subprocess.run([os.path.join(ROOT, "hobot_stereonet_amd/csrc/compat/build/node_harness"), "--bench", td + "/m.snw", td + "/s.bin",
                    str(w), str(h), str(frames)], capture_output=True, text=True, env=env)
sys.stderr.write("\n".join(l for l in r.stderr.splitlines() if "node stats" in l) + "\n")
print([l for l in r.stdout.splitlines() if l.startswith("{")][-1] if r.returncode == 0 else r.stderr[-800:])
