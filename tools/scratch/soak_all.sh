mkdir -p gpurun_out/soak
for f in fuzz_camera fuzz_misc fuzz_lidar fuzz_dense fuzz_pooling fuzz_heads; do
  timeout -k 10 260 python tests/soak/$f.py ${SOAK_SECONDS:-180} ${SOAK_SEED:-101} > gpurun_out/soak/$f.log 2>&1; echo "$f rc=$? $(tail -1 gpurun_out/soak/$f.log | cut -c1-150)"
done
timeout -k 10 300 python tools/scratch/long_run.py 2>&1 | tail -3 | cut -c1-200
