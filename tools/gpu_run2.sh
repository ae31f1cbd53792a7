#!/bin/bash
# GPU batch 2 (round 2): full -m gpu suite + full bench.py line (per-shape breakdown, train_step, res128)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80 > gpurun_out/r02_pytest2.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench2.json 2> gpurun_out/r02_bench2.err
tail -5 gpurun_out/r02_pytest2.log; grep -h "vs\|peak HBM\|losses" gpurun_out/r02_pytest2.log | head -30; cat gpurun_out/r02_bench2.json | cut -c1-300; tail -5 gpurun_out/r02_bench2.err
