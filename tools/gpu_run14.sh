#!/bin/bash
# GPU batch 14 (round 2): Winograd conv: register-ring weights (variant 200) vs LDS weight slot (0), epilogue prefetch
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest14a.log 2>&1
tail -4 gpurun_out/r02_pytest14a.log
MD_WINO_VARIANT=200 timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest14b.log 2>&1
tail -4 gpurun_out/r02_pytest14b.log
timeout 600 python tools/bench_wino.py --variants 0,200,1,16,201,216 --out gpurun_out/r02_wino_micro14.json > gpurun_out/r02_wino_micro14.log 2>&1
cat gpurun_out/r02_wino_micro14.log | cut -c1-200
