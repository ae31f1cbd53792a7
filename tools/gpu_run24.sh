#!/bin/bash
# GPU batch 24 (round 2): re-run of the bench-contract test after its expectation was updated to the Winograd kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q > gpurun_out/r02_pytest_gpu_cli_rerun.log 2>&1
tail -4 gpurun_out/r02_pytest_gpu_cli_rerun.log
