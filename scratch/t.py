import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from hobot_stereonet_amd import api, weights, synth
w, h, d = int(sys.argv[1]), int(sys.argv[2]), 96
weights.save_snw("/tmp/m.snw", weights.synthetic(0), w, h, d)
with api.StereoNetHIP("/tmp/m.snw", precision=api.PREC_F16) as e:
    x = synth.model_input_i8(w, h, d, 1)
    disp, raw = e.infer(x)
    print("ok", disp.mean())
