cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for shape in cfg2 cfg5; do for fl in "" "--flush"; do
timeout -k 10 200 python tools/ab_libs.py tools/scratch/libmmt_base.so mm_training_amd/libmmt_hip.so --rounds 8 --shape $shape --env-ab "$1" $fl 2> gpurun_out/envab.err | tee -a gpurun_out/envab.jsonl || { tail -5 gpurun_out/envab.err; exit 1; }
done; done
