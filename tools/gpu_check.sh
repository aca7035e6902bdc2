#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/gpu_check.sh [ab]'): the GPU test suite, then (with "ab") an
# interleaved A/B of a baseline build of the library against the current one and a chunk-size sweep of the
# forward.  A step that was killed by its timeout ends the script: no further GPU step is started after it.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out; mkdir -p $out
step() {   # step <seconds> <log> <cmd...>
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "[gpu_check] $* -> rc=$rc ($(date +%T))"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_check] timed out: stopping"; tail -5 "$log"; exit $rc; fi
  return $rc
}
step 800 $out/pytest_gpu.log python -m pytest tests -m gpu -x -q --durations=12
prc=$?
tail -25 $out/pytest_gpu.log
if [ "$1" = "ab" ] && [ -f mm_training_amd/libmmt_base.so ]; then
  step 150 $out/ab_cfg2.json python tools/ab_libs.py mm_training_amd/libmmt_base.so mm_training_amd/libmmt_hip.so --rounds 12
  cat $out/ab_cfg2.json
  step 150 $out/ab_cfg5.json python tools/ab_libs.py mm_training_amd/libmmt_base.so mm_training_amd/libmmt_hip.so --rounds 8 --shape cfg5
  cat $out/ab_cfg5.json
  # forward chunk sizes 232 / 312 / 400 / 464 (library default at cfg2) / 512: flags = 3 | (n/4 << 8)
  step 200 $out/kbench_chunks_cfg2.json python tools/kbench.py --shape cfg2 --algos 3,14851,19971,25603,29699,32771
  python - <<'EOF'
import json
t = open("gpurun_out/kbench_chunks_cfg2.json").read()
d = json.loads(t[t.index("{\n"):])
print("chunk sweep (interleaved, ms):", d.get("ab_interleaved_ms"))
print("alternating:", d.get("alternating"))
EOF
fi
exit $prc
