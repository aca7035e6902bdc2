#!/bin/bash
# Run ON THE GPU BOX after tools/collect_profiles.sh: the bench lines (no profiler attached) of every BASELINE
# config and the per-kernel micro-benchmarks whose JSON is committed under profiles/ (copy gpurun_out/final/* to profiles/r03_*).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/final; rm -rf $out; mkdir -p $out
run() { local log=$1; shift; timeout -k 10 300 "$@" > "$log" 2> "$log.err"; local rc=$?; echo "[final] $* -> rc=$rc ($(date +%T))"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc; return 0; }
run $out/bench_train_cfg4.jsonl python bench.py
run $out/bench_train_cfg4_ray_backward.jsonl python bench.py --lift-splat-backward ray --no-cpu-baseline --no-hotpath-leg
run $out/bench_train_cfg4_unfused.jsonl python bench.py --unfused --no-cpu-baseline
run $out/bench_train_cfg4_geom_form.jsonl python bench.py --geom-form --no-cpu-baseline --no-hotpath-leg
run $out/bench_train_cfg4_calibration_ids.jsonl python bench.py --calibration-ids --no-cpu-baseline --no-hotpath-leg
for c in cfg2 cfg3 cfg5; do run $out/bench_train_$c.jsonl python bench.py --config $c --no-cpu-baseline; done
# the step on ONE stream (no weight-gradient stream, task heads on the caller's stream): the A/B of ops/conv_overlap.py + the head streams
run $out/bench_train_cfg4_single_stream.jsonl python bench.py --conv-overlap off --head-streams 0 --no-cpu-baseline --no-hotpath-leg
run $out/bench_train_cfg5_single_stream.jsonl python bench.py --config cfg5 --conv-overlap off --head-streams 0 --no-cpu-baseline --no-hotpath-leg
# which layers the dense part of the step belongs to, and MIOpen on the dominant shapes in both layouts (full find)
run $out/conv_shapes_cfg4.txt python tools/conv_shapes.py cfg4 3
run $out/kbench_conv_layout.jsonl python tools/kbench_conv_layout.py
# what the gradient exchange costs ONE rank before any communication (one-rank RCCL group): native reducer, torch DDP, none
for m in native ddp plain; do MASTER_PORT=$((29800 + RANDOM % 150)) run $out/exchange_tax_$m.txt python tools/scratch/ddp_tax.py $m 30; done
# multi-stream soak: every (weight-gradient stream, head streams, amp) variant of the tiny step, 400 steps each, fresh processes
run $out/soak_streams.txt python tools/scratch/soak_streams.py 400
run $out/kbench_fused.txt env KBF_EXTRA=1 python tools/kbench_fused.py
run $out/kbench_camera_cfg4.json python tools/kbench_camera.py --shape cfg4
run $out/kbench_camera_cfg5_bf16.json python tools/kbench_camera.py --shape cfg5 --dtype bf16
run $out/kbench_camera_aim.json python tools/kbench_camera.py --shape aim
run $out/kbench_dcn.json python tools/kbench_dcn.py
run $out/kbench_lidar_and_producers.json python tools/kbench_lidar.py
run $out/ubench_atomic_rows.txt tools/ubench/atomic_rows
run $out/ubench_atomic_scatter.txt tools/ubench/atomic_scatter
grep -h '^{' $out/bench_*.jsonl | cut -c1-160
# what THIS run produced (gpurun merges into a local directory that may still hold files of earlier rounds: tools/adopt_profiles.sh
# copies only the names listed here)
ls $out | grep -v '\.err$' > $out/MANIFEST
