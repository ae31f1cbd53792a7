#!/bin/bash
# GPU batch 31 (round 2): Winograd kernel tests incl. the bit-exact weight-tile layout test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s > gpurun_out/r02_pytest31.log 2>&1
tail -4 gpurun_out/r02_pytest31.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest31.log | head
