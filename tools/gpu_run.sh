#!/bin/bash
# The ONE parametrised launcher for a gpurun call (replaces the numbered one-off scripts of rounds 1-2):
#   gpurun --timeout 900 -- 'tools/gpu_run.sh <name> "<step>" ["<step>" ...]'
# every step is a shell command run from the repo root under its own `timeout` (STEP_TIMEOUT seconds, default 600),
# stdout+stderr appended to gpurun_out/<name>.log, one status line per step printed to the call's tail.
# Profiling batches: tools/gpu_profile.sh (sampling step), tools/gpu_profile_train.sh (training step).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
name=$1; shift
mkdir -p gpurun_out
log=gpurun_out/$name.log
: > "$log"
for step in "$@"; do
  echo "=== $step" >> "$log"
  t0=$(date +%s)
  timeout "${STEP_TIMEOUT:-600}" bash -c "$step" >> "$log" 2>&1
  rc=$?
  echo "[gpu_run $name] rc=$rc $(( $(date +%s) - t0 ))s: $step"
done
tail -n "${TAIL_LINES:-40}" "$log" | cut -c1-300
