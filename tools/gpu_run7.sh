#!/bin/bash
# GPU batch 7 (round 2): A/B of the fused-operand transform schedule (alternating wave groups), dmtet normals tests
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
MD_BF_VARIANT=124 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "fused" > gpurun_out/r02_pytest7.log 2>&1
python -m pytest tests/test_gpu_dmtet.py -m gpu -q -s >> gpurun_out/r02_pytest7.log 2>&1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode"
$B > gpurun_out/r02_bench7_default.json 2> gpurun_out/r02_bench7.err
MD_BF_VARIANT=124 $B > gpurun_out/r02_bench7_alt.json 2>> gpurun_out/r02_bench7.err
$B > gpurun_out/r02_bench7_default2.json 2>> gpurun_out/r02_bench7.err
grep -h "passed\|failed" gpurun_out/r02_pytest7.log
for f in gpurun_out/r02_bench7_*.json; do echo $f; cut -c1-190 $f; done
