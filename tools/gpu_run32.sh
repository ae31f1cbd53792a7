#!/bin/bash
# GPU batch 32 (round 2): training-mode ResnetBlock through the Winograd path vs autograd through the oracle
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -s -k "winograd or dropout" > gpurun_out/r02_pytest32.log 2>&1
tail -6 gpurun_out/r02_pytest32.log; grep -h "^FAILED\|^ERROR\|Error" gpurun_out/r02_pytest32.log | head
