#!/bin/bash
# Wall time + ms/step of the default bench under different MIOpen find settings.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/miopen_modes
run() { tag=$1; shift; s=$(date +%s.%N); env "$@" timeout 600 python3 bench.py --steps 20 --warmup 8 > gpurun_out/miopen_modes/$tag.log 2>&1; e=$(date +%s.%N);
  echo "$tag wall=$(echo "$e - $s" | bc) $(grep '^{' gpurun_out/miopen_modes/$tag.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])')"; }
run A_default X=1
run B_nonaive MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
run C_mode3 MIOPEN_FIND_MODE=3 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
