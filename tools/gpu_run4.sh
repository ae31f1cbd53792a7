#!/bin/bash
# GPU batch 4 (round 2): full -m gpu suite + bench A/B: fused GroupNorm operand (half-item schedule) / fused attention / pipelined GEMMs
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r02_pytest4.log 2>&1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode"
$B > gpurun_out/r02_bench4_gn1_attn1.json 2> gpurun_out/r02_bench4.err
MD_FUSE_GN_APPLY=0 $B > gpurun_out/r02_bench4_gn0_attn1.json 2>> gpurun_out/r02_bench4.err
MD_FUSE_GN_APPLY=0 MD_FUSE_ATTN=0 $B > gpurun_out/r02_bench4_gn0_attn0.json 2>> gpurun_out/r02_bench4.err
tail -3 gpurun_out/r02_pytest4.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest4.log | head -20
for f in gpurun_out/r02_bench4_*.json; do echo $f; cut -c1-190 $f; done; tail -3 gpurun_out/r02_bench4.err
