#!/bin/bash
# GPU batch 5 (round 2): full -m gpu suite, the full default bench line, rocprofv3 kernel trace + PMC passes of the sampling step
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r02_pytest5.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench5.json 2> gpurun_out/r02_bench5.err
bash tools/gpu_profile.sh > gpurun_out/r02_profile5.log 2>&1
tail -3 gpurun_out/r02_pytest5.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest5.log | head -20
cut -c1-260 gpurun_out/r02_bench5.json; tail -3 gpurun_out/r02_bench5.err; head -12 gpurun_out/prof/kt.summary.txt
