#!/bin/bash
# GPU batch 20 (round 2): md_wino_prep with non-temporal stores of T; smoke()
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_wino.py --variants 0,0 --shapes 128:128:64:8,256:128:64:8,128:128:32:8 > gpurun_out/r02_wino_micro30.log 2>&1
cat gpurun_out/r02_wino_micro30.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
