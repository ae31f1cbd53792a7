#!/bin/bash
# GPU batch 9 (round 2): edge-case tests added after the evidence run, smoke(), C host
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dmtet.py tests/test_gpu_unet.py tests/test_gpu_kernels.py -m gpu -q -s > gpurun_out/r02_pytest9.log 2>&1
python __graft_entry__.py > gpurun_out/r02_smoke9.log 2>&1
tail -3 gpurun_out/r02_pytest9.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest9.log | head; tail -2 gpurun_out/r02_smoke9.log
