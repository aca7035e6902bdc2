#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): rocprofv3 kernel statistics of the default bench
# (train, BASELINE configs[3]) and of the isolated hot path (fp32 at the cfg2/cfg4 camera shape, bf16 at cfg5), plus the
# three PMC passes behind bench.py's roofline.traffic for each of them.  Summaries land in gpurun_out/profiles_new/
# (copy the ones to be judged into profiles/ as rNN_*: tools/adopt_profiles.sh).  One counter group per pass; kernel-trace/stats only in their own
# runs.  The program itself follows `--` (python3 bench.py), never a wrapper.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
# Optional arguments pick groups (stats stats1 pmc4 pmc5 pmc2 pmc5bf16; default: all, into an emptied directory).
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new; [ $# -eq 0 ] && rm -rf $out; mkdir -p $out
want() { [ -z "$GROUPS_WANTED" ] && return 0; case " $GROUPS_WANTED " in *" $1 "*) return 0;; esac; return 1; }
GROUPS_WANTED="$*"
raw=/tmp/mmt_prof; rm -rf $raw
# (a pass that stops writing for 7 minutes gets the whole call killed: 200 s is several times what a pass takes)
step() { local log=$1; shift; timeout -k 10 200 "$@" > "$log" 2>&1; local rc=$?; echo "[profiles] $(basename $log) rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then tail -5 "$log"; exit $rc; fi; }
stats() {  # stats <tag> <bench args...>
  local tag=$1; shift
  step $out/bench_${tag}_under_rocprof.log rocprofv3 --kernel-trace --stats --output-format csv -d $raw/$tag -o $tag -- python3 bench.py "$@"
  find $raw/$tag -name "*kernel_stats.csv" -exec cp {} $out/bench_${tag}_kernel_stats.csv \;
}
pmc() {    # pmc <tag> <bench args...>
  local tag=$1; shift
  local i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum"; do
    i=$((i+1))
    step $out/pmc_${tag}_$i.log rocprofv3 --pmc $ctrs --output-format csv -d $raw/pmc_${tag}_$i -o pmc -- python3 bench.py "$@"
  done
  python3 tools/aggregate_pmc.py "python bench.py $*" $raw/pmc_${tag}_1 $raw/pmc_${tag}_2 $raw/pmc_${tag}_3 > $out/pmc_$tag.json
}
if want stats; then
stats train_cfg4 --steps 20 --warmup 8 --no-cpu-baseline
stats train_cfg5 --config cfg5 --steps 12 --warmup 6 --no-cpu-baseline --no-hotpath-leg
stats hotpath_cfg2 --mode hotpath --config cfg2 --steps 50 --warmup 10 --no-cpu-baseline
stats hotpath_cfg5_bf16 --mode hotpath --config cfg5 --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline
fi
# the same step on ONE stream (no weight-gradient stream, task heads on the caller's stream): kernels that overlap are each charged the
# time they share the card, so per-kernel durations are read from this one
want stats1 && stats train_cfg4_single_stream --steps 20 --warmup 8 --no-cpu-baseline --conv-overlap off --head-streams 0
# (cfg4: 16 launches, the later 8 in the steady state of the exclusive-cell cache; cfg5's forward has no cache: 5 launches)
want pmc4 && pmc cfg4 --steps 8 --warmup 8 --no-cpu-baseline --no-hotpath-leg
want pmc5 && pmc cfg5 --config cfg5 --steps 3 --warmup 2 --no-cpu-baseline --no-hotpath-leg
want pmc2 && pmc cfg2 --mode hotpath --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline
want pmc5bf16 && pmc cfg5_bf16 --mode hotpath --config cfg5 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline
ls $out | grep -E '^(bench_.*(kernel_stats\.csv|under_rocprof\.log)|pmc_[a-z0-9_]*\.json)$' > $out/MANIFEST     # (see tools/adopt_profiles.sh)
grep -h '^{' $out/bench_*_under_rocprof.log | cut -c1-300
ls -la $out
