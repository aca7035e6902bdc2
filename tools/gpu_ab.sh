#!/bin/bash
# Run ON THE GPU BOX: quick parity tests of the touched kernels, interleaved A/B against the baseline build
# (mm_training_amd/libmmt_base.so) and per-kernel durations of the isolated hot path under rocprofv3.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_geometry_gpu.py tests/test_voxel_pooling_gpu.py tests/test_train_step_gpu.py -m gpu -x -q > $out/pytest_quick.log 2>&1
rc=$?; tail -4 $out/pytest_quick.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
# baseline build for the A/B: `cp mm_training_amd/libmmt_hip.so mm_training_amd/libmmt_base.so` BEFORE editing a kernel
# (the .so is git-ignored but travels with the gpurun snapshot)
[ -f mm_training_amd/libmmt_base.so ] && for shape in cfg2 cfg5; do
  timeout -k 10 150 python tools/ab_libs.py mm_training_amd/libmmt_base.so mm_training_amd/libmmt_hip.so --rounds 10 --shape $shape > $out/ab_$shape.json 2> $out/ab_$shape.err || { tail -5 $out/ab_$shape.err; exit 1; }
  cat $out/ab_$shape.json
done
raw=/tmp/mmt_prof; rm -rf $raw
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/hot -o hot -- python3 bench.py --mode hotpath --steps 50 --warmup 10 --no-cpu-baseline > $out/hot_under_rocprof.log 2>&1 || { tail -5 $out/hot_under_rocprof.log; exit 1; }
find $raw/hot -name "*kernel_stats.csv" -exec cp {} $out/hot_kernel_stats.csv \;
head -8 $out/hot_kernel_stats.csv | cut -c1-220
