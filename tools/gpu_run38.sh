#!/bin/bash
# GPU batch 38 (round 2): res128 B=2 parity and the training-mode Winograd block with the two-phase operand pass as default
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 110 python -m pytest tests/test_gpu_graded.py tests/test_gpu_backward.py -m gpu -q -s -k "res128 or winograd" > gpurun_out/r02_pytest38.log 2>&1
tail -5 gpurun_out/r02_pytest38.log | cut -c1-220
