#!/bin/bash
# GPU batch 6 (round 2): new/changed tests, NIN tile A/B, 999-step parity for 3 seeds, training-step kernel trace
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_graded.py tests/test_gpu_train.py tests/test_gpu_unet.py -m gpu -q -s > gpurun_out/r02_pytest6.log 2>&1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode"
$B > gpurun_out/r02_bench6_n128.json 2> gpurun_out/r02_bench6.err
MD_NIN_N128=0 $B > gpurun_out/r02_bench6_n256.json 2>> gpurun_out/r02_bench6.err
python tests/longrun_parity.py --steps 999 --seeds 42,43,44 --out gpurun_out/r02_longrun_999step_3seeds.json > gpurun_out/r02_longrun.log 2>&1
bash tools/gpu_profile_train.sh > gpurun_out/r02_profile_train.log 2>&1
tail -3 gpurun_out/r02_pytest6.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest6.log | head
for f in gpurun_out/r02_bench6_*.json; do echo $f; cut -c1-190 $f; done
tail -2 gpurun_out/r02_longrun.log | cut -c1-300
