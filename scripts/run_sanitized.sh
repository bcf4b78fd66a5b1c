#!/bin/bash
# CPU test suite of the checker (oracle/) and the host C++ mirror (compat/) under AddressSanitizer +
# UndefinedBehaviorSanitizer: both libraries are rebuilt with -fsanitize=address,undefined and loaded into the test
# process with the sanitizer runtimes preloaded.  The HIP product library is not instrumented (device code).
#   bash scripts/run_sanitized.sh [pytest args]
set -eu
cd "$(dirname "$0")/.."
make -C oracle -s asan
make -C hobot_stereonet_amd/csrc/compat -s asan
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export SN_SANITIZE=1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD="$ASAN:$UBSAN" python -m pytest -q -m "not gpu" -p no:cacheprovider \
  tests/test_oracle_preprocess.py tests/test_oracle_network.py tests/test_host_mirror.py tests/test_filelist.py tests/test_render_twin.py "$@"
