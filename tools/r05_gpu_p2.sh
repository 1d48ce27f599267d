#!/bin/bash
# a**2 as libm's pow on the device: the campaign (3-D differences from the oracle should be gone), then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > gpurun_out/p2_first.log 2>&1 || { tail -30 gpurun_out/p2_first.log; exit 1; }
timeout -k 10 700 python tools/mindist_campaign.py 3000 2>/dev/null | tail -12 > gpurun_out/mindist_campaign.txt; echo "campaign rc=$?"; cut -c1-700 gpurun_out/mindist_campaign.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/p2_tests.log 2>&1; rc=$?
tail -12 gpurun_out/p2_tests.log
exit $rc
