#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of bench.py, CSV output,
# summarised into gpurun_out/prof/*.txt (raw traces are deleted to stay under the 64 MiB copy-back cap).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
STEPS=${STEPS:-3}
BASE_ARGS="--no-res128 --no-train-step --no-fast-mode"   # the sampling step only (configs[1])
run() {  # name, rocprof args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 300 rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o $name -- python $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline $BASE_ARGS $BENCH_ARGS > $OUT/$name.log 2>&1
  python $R/tools/prof_summary.py /tmp/rp_$name $OUT/$name.summary.txt
  find /tmp/rp_$name -name "*kernel_stats.csv" -exec cp {} $OUT/$name.kernel_stats.csv \;
  tail -2 $OUT/$name.log
}
run kt --kernel-trace --stats
if [ -z "$SKIP_PMC" ]; then
STEPS=1 BENCH_ARGS="--no-kernel-events" run pmc_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
STEPS=1 BENCH_ARGS="--no-kernel-events" run pmc_fetch --kernel-trace --pmc FETCH_SIZE
STEPS=1 BENCH_ARGS="--no-kernel-events" run pmc_write --kernel-trace --pmc WRITE_SIZE
fi
head -30 $OUT/kt.summary.txt
