#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/bn_prof; rm -rf /tmp/bnp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnp -o t -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/bn_prof/log.txt 2>&1
find /tmp/bnp -name "*kernel_stats.csv" -exec cp {} gpurun_out/bn_prof/stats.csv \;
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/bn_prof/stats.csv')))
steps=14
for r in rows:
    n=r['Name']
    if 'bn_' in n or 'BatchNorm' in n or 'threshold' in n or 'clamp' in n or 'CUDAFunctor_add<float>' in n:
        print(f"{int(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step {int(r['Calls'])/steps:6.1f}/step avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f}  {n[:100]}")
PY
