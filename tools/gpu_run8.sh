#!/bin/bash
# GPU batch 8 (round 2): full -m gpu suite, the default bench line, rocprofv3 kernel trace + PMC passes (final build of the round)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r02_pytest8.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench8.json 2> gpurun_out/r02_bench8.err
bash tools/gpu_profile.sh > gpurun_out/r02_profile8.log 2>&1
tail -3 gpurun_out/r02_pytest8.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest8.log | head -20
cut -c1-260 gpurun_out/r02_bench8.json; tail -3 gpurun_out/r02_bench8.err; head -8 gpurun_out/prof/kt.summary.txt
