#!/bin/bash
# GPU batch 3 (round 2): full -m gpu suite (no -x) + bench with / without the fused GroupNorm operand loader
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r02_pytest3.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode > gpurun_out/r02_bench3_fused.json 2> gpurun_out/r02_bench3.err
MD_FUSE_GN_APPLY=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode > gpurun_out/r02_bench3_unfused.json 2>> gpurun_out/r02_bench3.err
tail -3 gpurun_out/r02_pytest3.log; grep -h "FAILED\|Error" gpurun_out/r02_pytest3.log | head -20; grep -h "vs \|peak HBM" gpurun_out/r02_pytest3.log | head -40; cut -c1-200 gpurun_out/r02_bench3_fused.json; cut -c1-200 gpurun_out/r02_bench3_unfused.json; tail -3 gpurun_out/r02_bench3.err
