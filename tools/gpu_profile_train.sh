#!/bin/bash
# Kernel-trace of the res64 training step (tools/train_step_bench.py) on the GPU box; summary in gpurun_out/prof/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_train -o train -- python $R/tools/train_step_bench.py --steps 1 --warmup 1 > $OUT/train.log 2>&1
python $R/tools/prof_summary.py /tmp/rp_train $OUT/train.summary.txt
find /tmp/rp_train -name "*kernel_stats.csv" -exec cp {} $OUT/train.kernel_stats.csv \;
tail -1 $OUT/train.log
head -40 $OUT/train.summary.txt
