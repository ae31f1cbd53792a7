#!/bin/bash
# GPU batch 18 (round 2): Winograd conv: halo through registers (variant 300)
export TMPDIR=/tmp
mkdir -p gpurun_out
MD_WINO_VARIANT=300 timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest18.log 2>&1
tail -3 gpurun_out/r02_pytest18.log
timeout 600 python tools/bench_wino.py --variants 0,200,300,301,316 --out gpurun_out/r02_wino_micro22.json > gpurun_out/r02_wino_micro22.log 2>&1
cat gpurun_out/r02_wino_micro22.log | cut -c1-200
