#!/bin/bash
# Builds an A/B copy of the library with extra compiler flags: scripts/build_variant.sh NAME -DSN_X=1 ...  -> scripts/build/libsn_NAME.so
NAME=$1; shift
mkdir -p scripts/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I include -ldl -lpthread "$@" \
  -o scripts/build/libsn_$NAME.so hobot_stereonet_amd/csrc/stereonet_hip.hip hobot_stereonet_amd/csrc/sn_mgpu.hip
