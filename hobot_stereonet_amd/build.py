"""Builds libstereonet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m hobot_stereonet_amd.build [--force]
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
DEPS = SOURCES + [os.path.join(CSRC, "sn_kernels.hpp"), os.path.join(CSRC, "sn_stream_block.hpp"), os.path.join(CSRC, "sn_tower_f32.hpp"), os.path.join(CSRC, "sn_agg_dma.hpp"), os.path.join(ROOT, "include", "stereonet_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-I", os.path.join(ROOT, "include"), "-ldl", "-lpthread"]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        cmd = [HIPCC] + FLAGS + ["-o", LIB] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
