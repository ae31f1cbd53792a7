#!/bin/bash
# GPU batch 33 (round 2): kernel trace + SQ counter pass of the sampling step on the last build
export TMPDIR=/tmp
mkdir -p gpurun_out
SKIP_PMC=1 bash tools/gpu_profile.sh > gpurun_out/r02_profile33.log 2>&1
R=$(pwd); OUT=$R/gpurun_out/prof
cd /tmp
rm -rf /tmp/rp_pmc_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/rp_pmc_sq -o pmc_sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-res128 --no-train-step --no-fast-mode --no-kernel-events > $OUT/pmc_sq.log 2>&1
python $R/tools/prof_summary.py /tmp/rp_pmc_sq $OUT/pmc_sq.summary.txt
cd $R
head -8 $OUT/kt.summary.txt | cut -c1-170; grep "md_conv3_wino" $OUT/pmc_sq.summary.txt | cut -c1-400
