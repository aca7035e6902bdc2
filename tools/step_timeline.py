#!/usr/bin/env python3
"""Where one training step's wall time goes, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o tl -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-hotpath-leg
    python3 tools/step_timeline.py DIR/**/tl_kernel_trace.csv
Steps are cut at the optimizer's update kernel (opt_adamw, dp/optim.py).  For the last-but-one step: wall time, the time at least one
kernel was running on ANY queue (union of the intervals), kernels and busy time per queue, the idle time of the device (no kernel on
any queue) split by the length of the hole, and the kernels the longest holes come BEFORE (the consumer that was late)."""
import csv
import sys
from collections import Counter, defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["t0"], r["t1"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["t0"])
    qcol = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    cuts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("opt_adamw") or "opt_adamw" in r["Kernel_Name"][:40]]
    if len(cuts) < 3:
        sys.exit("fewer than three optimizer updates in the trace")
    lo, hi = cuts[-3] + 1, cuts[-2] + 1
    step = rows[lo:hi]
    t_begin, t_end = rows[cuts[-3]]["t1"], step[-1]["t1"]
    wall = t_end - t_begin
    # union of the intervals
    busy, cur0, cur1 = 0, None, None
    holes = []
    prev_end = t_begin
    for r in step:
        if r["t0"] > prev_end:
            holes.append((r["t0"] - prev_end, r))
        prev_end = max(prev_end, r["t1"])
    idle = sum(h for h, _ in holes)
    print("step: %d kernels, wall %.2f ms, device idle (no kernel on any queue) %.2f ms = %.1f %%" % (len(step), wall / 1e6, idle / 1e6, 100.0 * idle / wall))
    per_q = defaultdict(lambda: [0, 0])
    for r in step:
        per_q[r[qcol]][0] += 1
        per_q[r[qcol]][1] += r["t1"] - r["t0"]
    print("kernels / busy ms per queue:", {q: (n, round(t / 1e6, 2)) for q, (n, t) in sorted(per_q.items())})
    bins = [(0, 2000), (2000, 5000), (5000, 10000), (10000, 30000), (30000, 100000), (100000, 10**12)]
    for a, b in bins:
        hs = [h for h, _ in holes if a <= h < b]
        print("holes %6.0f-%-8.0f us: %5d, %7.3f ms" % (a / 1e3, b / 1e3 if b < 10**11 else float("inf"), len(hs), sum(hs) / 1e6))
    late = Counter()
    for h, r in holes:
        late[r["Kernel_Name"][:90]] += h
    print("idle time in front of (top 25):")
    for k, v in late.most_common(25):
        n = sum(1 for h, r in holes if r["Kernel_Name"][:90] == k)
        print("  %8.3f ms  %5d x  %s" % (v / 1e6, n, k))
    dur = Counter()
    cnt = Counter()
    for r in step:
        dur[r["Kernel_Name"][:90]] += r["t1"] - r["t0"]
        cnt[r["Kernel_Name"][:90]] += 1
    print("kernel time (top 25; sums over queues, so overlapping kernels count twice):")
    for k, v in dur.most_common(25):
        print("  %8.3f ms  %5d x  %s" % (v / 1e6, cnt[k], k))


if __name__ == "__main__":
    main()
