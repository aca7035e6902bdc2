#!/bin/bash
# Run ON THE GPU BOX: `bash tools/gpu_step.sh <name> <seconds> <cmd...>` runs one GPU step under its own timeout, logs
# to gpurun_out/<name>.log (+ .err) and refuses to start when an earlier step of the same call was killed by its
# timeout (marker file gpurun_out/.step_killed, removed by the first step of a fresh box since nothing persists there).
name=$1; secs=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ -f /tmp/.mmt_step_killed ]; then echo "[gpu_step] $name: skipped (an earlier step timed out)"; exit 124; fi
timeout -k 10 "$secs" "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err
rc=$?
echo "[gpu_step] $name -> rc=$rc ($(date +%T))"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then touch /tmp/.mmt_step_killed; tail -5 gpurun_out/$name.err; fi
exit $rc
