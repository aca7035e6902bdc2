#!/bin/bash
# Run ON THE GPU BOX at the final tree of round 6: the bench lines (no profiler attached) and the kernel micro-benchmarks committed
# under profiles/r06_* (tools/adopt_profiles.sh r06 copies what the MANIFEST lists).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/final; rm -rf $out; mkdir -p $out
run() { local log=$1; shift; timeout -k 10 300 "$@" > "$log" 2> "$log.err"; local rc=$?; echo "[final] $* -> rc=$rc ($(date +%T))"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc; return 0; }
run $out/bench_train_cfg4.jsonl python bench.py
run $out/bench_train_cfg4_calibration_ids.jsonl python bench.py --calibration-ids --no-cpu-baseline --no-hotpath-leg
run $out/bench_train_cfg5.jsonl python bench.py --config cfg5 --no-cpu-baseline
run $out/bench_train_aim.jsonl python bench.py --config aim --no-cpu-baseline --no-hotpath-leg
run $out/bench_train_cfg4_nuscenes_rig.jsonl python bench.py --rig nuscenes --no-cpu-baseline --no-hotpath-leg
run $out/kbench_camera_cfg4.json python tools/kbench_camera.py --shape cfg4
run $out/kbench_camera_cfg5_bf16.json python tools/kbench_camera.py --shape cfg5 --dtype bf16
run $out/kbench_camera_aim.json python tools/kbench_camera.py --shape aim
run $out/kbench_voxelize.json python tools/kbench_voxelize.py
run $out/kbench_voxelize_fused_launch.json env MMT_VOX_FUSED=1 python tools/kbench_voxelize.py
run $out/kbench_lidar_and_producers.json python tools/kbench_lidar.py
grep -h '^{' $out/bench_*.jsonl | cut -c1-200
ls $out | grep -v '\.err$' > $out/MANIFEST
