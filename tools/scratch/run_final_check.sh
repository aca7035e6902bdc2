cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python bench.py 2> gpurun_out/bench_final.err | tee gpurun_out/bench_final.jsonl | cut -c1-300
