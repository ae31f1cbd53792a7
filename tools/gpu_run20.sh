#!/bin/bash
# GPU batch 20 (round 2): Winograd conv: cost of the statistics / residual parts of the epilogue
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_wino.py --variants 0,0 --shapes 128:128:64:8 > gpurun_out/r02_wino_micro28.log 2>&1
python tools/bench_wino.py --variants 0,0 --shapes 128:128:64:8 --no-stats >> gpurun_out/r02_wino_micro28.log 2>&1
python tools/bench_wino.py --variants 0,0 --shapes 128:128:64:8 --no-stats --no-res >> gpurun_out/r02_wino_micro28.log 2>&1
cat gpurun_out/r02_wino_micro28.log | cut -c1-200
