# kernel statistics of the default bench only (one rocprofv3 pass) -> gpurun_out/profiles_new/bench_train_cfg4_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new; mkdir -p $out
raw=/tmp/mmt_prof1; rm -rf $raw
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/t -o t -- python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline > $out/bench_train_cfg4_under_rocprof.log 2>&1
find $raw/t -name "*kernel_stats.csv" -exec cp {} $out/bench_train_cfg4_kernel_stats.csv \;
grep -E "finalize|bn_" $out/bench_train_cfg4_kernel_stats.csv | cut -c1-160
