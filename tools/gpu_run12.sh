#!/bin/bash
# GPU batch 12 (round 2): Winograd conv path: kernel tests, U-Net parity with it enabled, same-box A/B on the bench
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -s -x > gpurun_out/r02_pytest12a.log 2>&1
tail -15 gpurun_out/r02_pytest12a.log
if grep -q "passed" gpurun_out/r02_pytest12a.log && ! grep -q "failed" gpurun_out/r02_pytest12a.log; then
  B="timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode"
  $B > gpurun_out/r02_bench12_on.json 2> gpurun_out/r02_bench12.err
  MD_WINO=0 $B > gpurun_out/r02_bench12_off.json 2>> gpurun_out/r02_bench12.err
  timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_graded.py -m gpu -q -s > gpurun_out/r02_pytest12b.log 2>&1
  tail -5 gpurun_out/r02_pytest12b.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest12b.log | head
  for f in gpurun_out/r02_bench12_*.json; do echo $f; cut -c1-190 $f; done
  tail -5 gpurun_out/r02_bench12.err
fi
