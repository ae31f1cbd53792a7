#!/bin/bash
# GPU batch 10 (round 2): deep-prefetch GEMM loop: correctness (kernel + unet tests) and same-box A/B on the bench
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_backward.py -m gpu -q -s > gpurun_out/r02_pytest10.log 2>&1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode"
$B > gpurun_out/r02_bench10_p4.json 2> gpurun_out/r02_bench10.err
MD_NIN_P1=1 $B > gpurun_out/r02_bench10_p1.json 2>> gpurun_out/r02_bench10.err
$B > gpurun_out/r02_bench10_p4b.json 2>> gpurun_out/r02_bench10.err
tail -3 gpurun_out/r02_pytest10.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest10.log | head
for f in gpurun_out/r02_bench10_*.json; do echo $f; cut -c1-190 $f; done
