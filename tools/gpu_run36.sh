#!/bin/bash
# GPU batch 36 (round 2): experimental two-phase operand pass md_wino_prep_v2: bit-identity + timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -k "prep_v2" > gpurun_out/r02_pytest36.log 2>&1
tail -4 gpurun_out/r02_pytest36.log | cut -c1-200
timeout 300 python tools/bench_wino.py --variants 0 --prep-v2 --shapes 128:128:64:8,256:128:64:8,128:128:32:8 > gpurun_out/r02_wino_micro36.log 2>&1
grep "prep" gpurun_out/r02_wino_micro36.log | cut -c1-200
