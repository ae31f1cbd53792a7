#!/bin/bash
# GPU batch 1 (round 2): full -m gpu suite + conv kernel A/B (stagger / early weight commit) + bench baseline
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r02_pytest1.log
python tools/bench_conv.py --cfgs C3_128_FAST,FAST_EC,C3_128_FAST_STAG,FAST_EC_STAG --stagger 40000 --shapes 0,1,2 --rounds 5 > gpurun_out/r02_conv_ab1.log 2>&1
python tools/bench_conv.py --cfgs C3_128_FAST,C3_128_FAST_STAG --stagger 100000 --shapes 0,1 --rounds 5 >> gpurun_out/r02_conv_ab1.log 2>&1
python tools/bench_conv.py --cfgs C3_128_FAST,FAST_EC,C3_128_FAST_STAG --stagger 40000 --shapes 0,1 --rounds 5 --no-residual >> gpurun_out/r02_conv_ab1.log 2>&1
python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline > gpurun_out/r02_bench_base.json 2> gpurun_out/r02_bench_base.err
MD_CONV_STAGGER=40000 python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline > gpurun_out/r02_bench_stag.json 2>> gpurun_out/r02_bench_base.err
tail -5 gpurun_out/r02_pytest1.log; cat gpurun_out/r02_conv_ab1.log | grep "shape B"; cat gpurun_out/r02_bench_base.json | cut -c1-400; cat gpurun_out/r02_bench_stag.json | cut -c1-400
