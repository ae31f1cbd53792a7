#!/bin/bash
# GPU batch 20 (round 2): Winograd conv: real-data ablations (what the halo / weight traffic costs when the operands keep their values)
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_wino.py --variants 0,0,2,4,6,22,16 --shapes 128:128:64:8,256:128:64:8 > gpurun_out/r02_wino_micro29.log 2>&1
cat gpurun_out/r02_wino_micro29.log | cut -c1-200
