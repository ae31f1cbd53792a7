#!/bin/bash
# GPU batch 29 (round 2): kernel trace of the training step of the last build (Winograd forward + data-gradient convs)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_profile_train.sh > gpurun_out/r02_profile_train29.log 2>&1
tail -32 gpurun_out/r02_profile_train29.log | cut -c1-170
