#!/bin/bash
# GPU batch 28 (round 2): default bench line of the last build
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
cut -c1-260 gpurun_out/r02_final_bench.json; tail -2 gpurun_out/r02_final_bench.err
