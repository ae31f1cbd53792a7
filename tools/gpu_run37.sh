#!/bin/bash
# GPU batch 37 (round 2): two-phase operand pass as the default: Winograd + U-Net parity tests, default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_unet.py -m gpu -q > gpurun_out/r02_pytest37.log 2>&1
tail -3 gpurun_out/r02_pytest37.log | cut -c1-200; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest37.log | head
timeout 600 python bench.py > gpurun_out/r02_final_bench_prep2.json 2> gpurun_out/r02_final_bench_prep2.err
cut -c1-240 gpurun_out/r02_final_bench_prep2.json
