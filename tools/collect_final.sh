#!/bin/bash
# Run ON THE GPU BOX after tools/collect_profiles.sh: the bench lines (no profiler attached) of every BASELINE
# config and the per-kernel micro-benchmarks whose JSON is committed under profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/final; rm -rf $out; mkdir -p $out
run() { local log=$1; shift; timeout -k 10 400 "$@" > "$log" 2> "$log.err"; local rc=$?; echo "[final] $* -> rc=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc; return 0; }
run $out/bench_train_cfg2.jsonl python bench.py
run $out/bench_hotpath_cfg2.jsonl python bench.py --mode hotpath --steps 50 --warmup 10
run $out/bench_train_cfg2_cached_plan.jsonl python bench.py --cached-plan --no-cpu-baseline
run $out/bench_train_cfg2_fused_lift_splat.jsonl python bench.py --fused-lift-splat --no-cpu-baseline
for c in cfg3 cfg4 cfg5; do run $out/bench_train_$c.jsonl python bench.py --config $c --no-cpu-baseline; done
KBENCH_VERIFY=1 run $out/kbench_voxel_pooling_cfg2_rig.json python tools/kbench.py --shape cfg2 --algos 3,67,4
KBENCH_VERIFY=1 run $out/kbench_voxel_pooling_cfg5.json python tools/kbench.py --shape cfg5 --algos 3
KBENCH_VERIFY=1 run $out/kbench_voxel_pooling_cfg1_full.json python tools/kbench.py --shape cfg1_full --algos 3
KBENCH_VERIFY=1 run $out/kbench_voxel_pooling_aim.json python tools/kbench.py --shape aim --algos 3
run $out/kbench_voxel_pooling_cfg2_uniform.json python tools/kbench.py --shape cfg2 --geometry uniform --algos 3
run $out/kbench_lidar_and_producers.json python tools/kbench_lidar.py
grep -h '^{' $out/bench_*.jsonl | cut -c1-160
