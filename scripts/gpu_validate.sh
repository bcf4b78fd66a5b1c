#!/bin/bash
# one gpurun call: the whole GPU test suite, then the EPE envelope table (scripts/epe_sensitivity.py) per configuration
TAG=${1:-r05}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/${TAG}_gputests.log
timeout 1400 python scripts/epe_sensitivity.py --config c2 > gpurun_out/${TAG}_epe_sensitivity_c2.txt 2> gpurun_out/${TAG}_epe_sens.err; tail -22 gpurun_out/${TAG}_epe_sensitivity_c2.txt
timeout 1400 python scripts/epe_sensitivity.py --config c5 > gpurun_out/${TAG}_epe_sensitivity_c5.txt 2>> gpurun_out/${TAG}_epe_sens.err; tail -22 gpurun_out/${TAG}_epe_sensitivity_c5.txt
