#!/bin/bash
# GPU batch 17 (round 2): Winograd conv: T with hi/lo planes side by side (128-byte halo rows)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest17a.log 2>&1
tail -3 gpurun_out/r02_pytest17a.log
MD_WINO_VARIANT=200 timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest17b.log 2>&1
tail -3 gpurun_out/r02_pytest17b.log
timeout 600 python tools/bench_wino.py --variants 0,200,201,1 --out gpurun_out/r02_wino_micro21.json > gpurun_out/r02_wino_micro21.log 2>&1
cat gpurun_out/r02_wino_micro21.log | cut -c1-200
