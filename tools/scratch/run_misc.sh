cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
# functional rehearsal of the N>1 launch path on the one GPU (ranks share cuda:0 -> gloo on every rank, see bench.choose_backend)
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/bench_2rank.log 2>&1; echo "2-rank rc=$?"; grep '^{' gpurun_out/bench_2rank.log | cut -c1-420; tail -3 gpurun_out/bench_2rank.log | cut -c1-300
