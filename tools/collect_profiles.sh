#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): rocprofv3 kernel statistics of the
# default bench (train, cfg2) and of the isolated hot path, plus the three PMC passes behind
# bench.py's roofline.traffic.  Summaries land in gpurun_out/profiles_new/ (copy the ones to be
# judged into profiles/).  One counter group per pass, kernel-trace/stats only in their own runs.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new; rm -rf $out; mkdir -p $out
raw=/tmp/mmt_prof; rm -rf $raw
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/train -o train -- python3 bench.py --steps 20 --warmup 8 > $out/bench_train_under_rocprof.log 2>&1
find $raw/train -name "*kernel_stats.csv" -exec cp {} $out/bench_train_cfg2_kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/hot -o hot -- python3 bench.py --mode hotpath --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_hotpath_under_rocprof.log 2>&1
find $raw/hot -name "*kernel_stats.csv" -exec cp {} $out/bench_hotpath_cfg2_kernel_stats.csv \;
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d $raw/pmc$i -o pmc -- python3 bench.py --mode hotpath --steps 5 --warmup 2 --no-cpu-baseline > $out/pmc$i.log 2>&1
done
python3 tools/aggregate_pmc.py $raw/pmc1 $raw/pmc2 $raw/pmc3 > $out/hotpath_cfg2_pmc.json
grep -h '^{' $out/bench_train_under_rocprof.log $out/bench_hotpath_under_rocprof.log | cut -c1-200
head -c 1500 $out/hotpath_cfg2_pmc.json
