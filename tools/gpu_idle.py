#!/usr/bin/env python3
"""GPU idle time inside the training step, from a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per dispatch):
    python tools/gpu_idle.py <kernel_trace.csv> [steps_to_skip_at_each_end]
Splits the trace at the optimizer's fused AdamW kernel (one per step), and for every complete step reports its span, the time
some kernel was running, and the idle remainder (gaps between dependent launches, host stalls)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [i for i, e in enumerate(ev) if "multi_tensor_apply_kernel" in e[2] and "Adam" in e[2]]
# one step may launch several AdamW chunks: keep the LAST of each cluster
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or ev[marks[i + 1]][0] - ev[marks[i]][1] > 5_000_000]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = ev[a + 1:b + 1]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    steps.append((t1 - t0, busy, len(seg)))
steps = steps[skip:len(steps) - skip] if len(steps) > 2 * skip else steps
for span, busy, n in steps:
    print("step: span %.2f ms, busy %.2f ms, idle %.2f ms (%.1f %%), %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, n))
if steps:
    sp = sum(s for s, _, _ in steps) / len(steps)
    bz = sum(b for _, b, _ in steps) / len(steps)
    print("mean over %d steps: span %.2f ms, busy %.2f ms, idle %.2f ms = %.1f %%" % (len(steps), sp / 1e6, bz / 1e6, (sp - bz) / 1e6, 100.0 * (sp - bz) / sp))
