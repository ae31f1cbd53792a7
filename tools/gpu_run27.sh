#!/bin/bash
# GPU batch 27 (round 2): final evidence of the last build: full GPU tests, PMC fetch/write passes (conv traffic key), default bench
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu_final2.log 2>&1
tail -3 gpurun_out/r02_pytest_gpu_final2.log; grep -h "^FAILED\|^ERROR" gpurun_out/r02_pytest_gpu_final2.log | head
OUT=$R/gpurun_out/prof2; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  rm -rf /tmp/rp_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp_$n -o $n -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode --no-kernel-events > $OUT/$n.log 2>&1
  python $R/tools/prof_summary.py /tmp/rp_$n $OUT/$n.summary.txt
done
cd $R
grep -h "md_conv3_wino\|md_wino_prep" $OUT/*.summary.txt | cut -c1-200
