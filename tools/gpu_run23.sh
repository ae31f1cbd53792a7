#!/bin/bash
# GPU batch 23 (round 2): final full GPU test suite + default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu_final.log 2>&1
tail -5 gpurun_out/r02_pytest_gpu_final.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest_gpu_final.log | head
timeout 900 python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
cut -c1-300 gpurun_out/r02_final_bench.json; tail -2 gpurun_out/r02_final_bench.err
