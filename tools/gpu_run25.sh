#!/bin/bash
# GPU batch 25 (round 2): Winograd forward convs in the training step: backward / train tests + train step timing A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_train.py -m gpu -q -s > gpurun_out/r02_pytest26.log 2>&1
tail -4 gpurun_out/r02_pytest26.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest26.log | head
for v in 1 0; do
MD_WINO_TRAIN_FWD=$v timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-res128 --no-fast-mode --no-kernel-events > gpurun_out/r02_bench26_$v.json 2> gpurun_out/r02_bench26.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_bench26_$v.json").read().strip().splitlines()[-1])
t=d["train_step"]; print("MD_WINO_TRAIN_FWD=$v", d["ms_per_step"], t["ms_per_step"], t["split_ms"], t["loss"], t["peak_hbm_gib"])
PY
done
tail -3 gpurun_out/r02_bench26.err
