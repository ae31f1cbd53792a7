#!/bin/bash
# GPU batch 30 (round 2): lazy packing of the direct weight tiles: full GPU tests + training step timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu_final3.log 2>&1
tail -3 gpurun_out/r02_pytest_gpu_final3.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest_gpu_final3.log | head
timeout 600 python tools/train_step_bench.py --steps 3 --warmup 2 > gpurun_out/r02_train30.json 2> gpurun_out/r02_train30.err
tail -1 gpurun_out/r02_train30.json | cut -c1-600
