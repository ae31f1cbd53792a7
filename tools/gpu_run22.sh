#!/bin/bash
# GPU batch 22 (round 2): final profile of the sampling step (kernel trace + PMC passes) and the 999-step parity run
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
bash tools/gpu_profile.sh > gpurun_out/r02_profile22.log 2>&1
tail -25 gpurun_out/r02_profile22.log | cut -c1-200
cd $R
timeout 900 python tests/longrun_parity.py --steps 999 --seeds 42,43 --out gpurun_out/r02_longrun22.json > gpurun_out/r02_longrun22.log 2>&1
tail -2 gpurun_out/r02_longrun22.log | cut -c1-600
