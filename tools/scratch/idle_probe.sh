#!/bin/bash
# GPU idle time inside the training step: kernel-trace timestamps of a short run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/idle; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/idle -o t -- python3 bench.py --steps 8 --warmup 6 --no-cpu-baseline > /tmp/idle_log.txt 2>&1
grep '^{' /tmp/idle_log.txt | cut -c75-170
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/idle/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: last 60 % of the trace
t0 = rows[int(len(rows) * 0.4)][0]
rows = [r for r in rows if r[0] >= t0]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
cur_end = rows[0][1]
for s, e, n in rows[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, n))
    cur_end = max(cur_end, e)
idle = sum(g for g, _ in gaps)
print(f"kernels {len(rows)}  span {span/1e6:.1f} ms  busy(sum of durations) {busy/1e6:.1f} ms  idle(gaps) {idle/1e6:.1f} ms = {100*idle/span:.1f} %")
import collections
big = sorted(gaps, reverse=True)[:8]
print("largest gaps (us, next kernel):", [(round(g/1e3,1), n[:50]) for g, n in big])
hist = collections.Counter(min(int(g/1000), 50)//5*5 for g, _ in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
PY
