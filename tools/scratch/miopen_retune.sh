#!/bin/bash
# Regenerate the MIOpen user find/perf DB for the cfg2 training step, then measure with it.
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/miopen_db2; mkdir -p $out
NONAIVE="MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0"
t0=$(date +%s)
env $NONAIVE MIOPEN_USER_DB_PATH=$out MIOPEN_FIND_MODE=1 MIOPEN_FIND_ENFORCE=4 timeout 900 python3 bench.py --miopen-tune --steps 10 --warmup 4 > $out/tune.log 2>&1
t1=$(date +%s); echo "tune wall $((t1-t0)) s"; grep '^{' $out/tune.log | cut -c1-160
for mode in 2 3; do
  t0=$(date +%s)
  env $NONAIVE MIOPEN_USER_DB_PATH=$out MIOPEN_FIND_MODE=$mode timeout 600 python3 bench.py --miopen-tune --steps 20 --warmup 8 > $out/mode$mode.log 2>&1
  t1=$(date +%s); echo "mode$mode wall $((t1-t0)) s"; grep '^{' $out/mode$mode.log | cut -c1-160
done
ls -la $out
