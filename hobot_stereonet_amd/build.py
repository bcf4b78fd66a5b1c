"""Builds libstereonet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m hobot_stereonet_amd.build [--force] [--diag]

--diag builds libstereonet_hip_diag.so instead: the same sources with -DSN_DIAGNOSTICS=1 (the precision-ablation switches
SN_ABLATE_W / SN_ABLATE_X that scripts/lowres_ablation.py drives; load it through STEREONET_HIP_LIB).  The shipping
library contains none of that code.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libstereonet_hip.so")
SOURCES = [os.path.join(CSRC, "stereonet_hip.hip"), os.path.join(CSRC, "sn_mgpu.hip")]
# every header under csrc/ (not the host mirror in csrc/compat, which has its own Makefile) + the C ABI
DEPS = SOURCES + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))) + [os.path.join(ROOT, "include", "stereonet_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-I", os.path.join(ROOT, "include"), "-ldl", "-lpthread"]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


LIB_DIAG = os.path.join(PKG, "libstereonet_hip_diag.so")


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        cmd = [HIPCC] + FLAGS + ["-o", LIB] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_diag(verbose: bool = False) -> str:
    cmd = [HIPCC] + FLAGS + ["-DSN_DIAGNOSTICS=1", "-o", LIB_DIAG] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_DIAG


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
