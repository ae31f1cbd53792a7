#!/bin/bash
# GPU batch 19 (round 2): Winograd conv, cleaned-up kernel (halo through registers): tests + bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_unet.py tests/test_gpu_graded.py -m gpu -q -s > gpurun_out/r02_pytest19.log 2>&1
tail -4 gpurun_out/r02_pytest19.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest19.log | head
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-fast-mode > gpurun_out/r02_bench19.json 2> gpurun_out/r02_bench19.err
cut -c1-260 gpurun_out/r02_bench19.json; tail -3 gpurun_out/r02_bench19.err
