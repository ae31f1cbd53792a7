#!/bin/bash
# GPU batch 34 (round 2): experimental F(4,3) Winograd kernels: correctness + micro-benchmark beside F(2,3)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -k "wino43" > gpurun_out/r02_pytest35.log 2>&1
tail -8 gpurun_out/r02_pytest35.log | cut -c1-200
timeout 300 python tools/bench_wino.py --variants 0,0 --f43 --shapes 128:128:64:8,256:128:64:8 > gpurun_out/r02_wino_micro35.log 2>&1
grep shape gpurun_out/r02_wino_micro35.log | cut -c1-200
