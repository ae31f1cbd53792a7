#!/bin/bash
# GPU batch 16 (round 2): Winograd conv: all halo pieces of a chunk in one step (variant 200)
export TMPDIR=/tmp
mkdir -p gpurun_out
MD_WINO_VARIANT=200 timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest16.log 2>&1
tail -3 gpurun_out/r02_pytest16.log
timeout 600 python tools/bench_wino.py --variants 0,200,201,216 --out gpurun_out/r02_wino_micro20.json > gpurun_out/r02_wino_micro20.log 2>&1
cat gpurun_out/r02_wino_micro20.log | cut -c1-200
