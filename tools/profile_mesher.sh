#!/bin/bash
# rocprofv3 kernel trace of the marching-tets launch (BASELINE configs[4], 32 meshes): gpurun_out/prof/mesher.summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_mesher
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_mesher -o mesher -- python -c "
import sys, json, torch
sys.path.insert(0, '$R')
import bench
print(json.dumps(bench.marching_tets_bench(torch.device('cuda:0'))))
" > $OUT/mesher.log 2>&1
python $R/tools/prof_summary.py /tmp/rp_mesher $OUT/mesher.summary.txt
tail -1 $OUT/mesher.log | cut -c1-400
grep "md_mt_\|kernel trace" $OUT/mesher.summary.txt
