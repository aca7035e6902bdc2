# usage (GPU box): bash tools/scratch/step_timeline.sh cfg4   -> gpurun_out/tl/step_timeline_cfg4.txt
cfg=${1:-cfg4}
mkdir -p gpurun_out/tl && export TMPDIR=/tmp
rm -rf gpurun_out/tl/$cfg
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$cfg -o tl -- python3 bench.py --config $cfg --steps 8 --warmup 5 --no-cpu-baseline --no-hotpath-leg > gpurun_out/tl/$cfg.log 2>&1
f=$(find gpurun_out/tl/$cfg -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f > gpurun_out/tl/step_timeline_$cfg.txt 2>&1
rm -rf gpurun_out/tl/$cfg
head -12 gpurun_out/tl/step_timeline_$cfg.txt
