#!/bin/bash
# GPU batch 11 (round 2): md_nin_f32 streaming shortcut NIN: correctness and same-box A/B on the bench
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_graded.py -m gpu -q -s > gpurun_out/r02_pytest11.log 2>&1
B="timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode"
$B > gpurun_out/r02_bench11_on.json 2> gpurun_out/r02_bench11.err
MD_NIN_STREAM=0 $B > gpurun_out/r02_bench11_off.json 2>> gpurun_out/r02_bench11.err
$B > gpurun_out/r02_bench11_on2.json 2>> gpurun_out/r02_bench11.err
tail -3 gpurun_out/r02_pytest11.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest11.log | head
for f in gpurun_out/r02_bench11_*.json; do echo $f; cut -c1-190 $f; done
