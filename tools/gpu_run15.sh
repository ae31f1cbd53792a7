#!/bin/bash
# GPU batch 15 (round 2): Winograd conv: is the halo DMA cost HBM latency (L2-resident halo ablation)?
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/bench_wino.py --variants 300,428,556,301 --shapes 128:128:64:8 --out gpurun_out/r02_wino_micro24.json > gpurun_out/r02_wino_micro24.log 2>&1
cat gpurun_out/r02_wino_micro24.log | cut -c1-200
