#!/bin/bash
# GPU batch 21 (round 2): Winograd data-gradient convs in the training backward: tests + train step timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q -s > gpurun_out/r02_pytest21.log 2>&1
tail -4 gpurun_out/r02_pytest21.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest21.log | head
timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-res128 --no-fast-mode --no-kernel-events > gpurun_out/r02_bench21.json 2> gpurun_out/r02_bench21.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench21.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["train_step"])[:900])
PY
tail -3 gpurun_out/r02_bench21.err
