#!/bin/bash
# GPU batch 13 (round 2): Winograd conv: new interleaved schedule correctness + microbenchmark with ablations
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest13.log 2>&1
tail -12 gpurun_out/r02_pytest13.log
timeout 600 python tools/bench_wino.py --out gpurun_out/r02_wino_micro13.json > gpurun_out/r02_wino_micro13.log 2>&1
cat gpurun_out/r02_wino_micro13.log | cut -c1-220
