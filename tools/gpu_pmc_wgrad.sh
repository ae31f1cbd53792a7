#!/bin/bash
# PMC passes over tools/bench_wgrad.py --only 0 (128->128 @ 64^3, the dominant wgrad shape); summaries in gpurun_out/prof/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 300 rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o $name -- python $R/tools/bench_wgrad.py --only 0 --iters 1 $BENCH_ARGS > $OUT/$name.log 2>&1
  python $R/tools/prof_summary.py /tmp/rp_$name $OUT/$name.summary.txt
  grep -E "wgrad_kernel|^#|^kernel" $OUT/$name.summary.txt | head -8
}
if [ -n "$CLOCK_ONLY" ]; then run wg_clk --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT; cat $OUT/wg_clk.summary.txt | cut -c1-200 | grep -A12 counters; exit 0; fi
run wg_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run wg_sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT
