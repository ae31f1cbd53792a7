#!/bin/bash
# Kernel-trace of res128 B=2 sampling steps (BASELINE config #4); summary in gpurun_out/prof/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_r128
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_r128 -o r128 -- python $R/tools/run_configs.py --only-res128 > $OUT/r128.log 2>&1
python $R/tools/prof_summary.py /tmp/rp_r128 $OUT/r128.summary.txt
tail -1 $OUT/r128.log
head -24 $OUT/r128.summary.txt
