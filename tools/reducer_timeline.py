#!/usr/bin/env python3
"""Timeline of the native gradient reducer inside one training step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o tax -- python3 tools/scratch/ddp_tax.py native 6
    python3 tools/reducer_timeline.py DIR/**/tax_kernel_trace.csv > profiles/rNN_reducer_timeline.txt
Per bucket of the LAST complete step: when its pack kernel (one multi-tensor copy per bucket, dp/reducer.py) started and ended, the RCCL
kernel that followed it (a one-rank group launches none or a copy), and where the backward pass ended (the first kernel of the gradient
clip's norm).  Times in microseconds relative to the step's first backward kernel; `queue` tells the streams apart."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["t0"], r["t1"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["t0"])
    name = lambda r: r["Kernel_Name"]
    is_opt = lambda r: "FusedAdam" in name(r) or "FusedOptimizer" in name(r)
    is_norm = lambda r: "LpNorm" in name(r) or "lpnorm" in name(r).lower()
    is_pack = lambda r: "multi_tensor_apply" in name(r) and ("Copy" in name(r) or "copy" in name(r))
    is_rccl = lambda r: "nccl" in name(r).lower() or "rccl" in name(r).lower()
    opt = [i for i, r in enumerate(rows) if is_opt(r)]
    # optimizer launches come in a burst per step: step boundaries = gaps of more than 5 ms between them
    ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or rows[opt[k + 1]]["t0"] - rows[i]["t1"] > 5_000_000]
    if len(ends) < 3:
        sys.exit("fewer than three steps in the trace")
    lo, hi = ends[-3] + 1, ends[-2] + 1                      # kernels of the last-but-one step (the last one may be cut by the exit)
    step = rows[lo:hi]
    norms = [r for r in step if is_norm(r)]
    first_norm = norms[0]["t0"] if norms else None
    packs = [r for r in step if is_pack(r) and (first_norm is None or r["t0"] < first_norm + 2_000_000)]
    # the backward pass starts after the loss: take the first weight-gradient kernel as the origin
    wrw = [r for r in step if "wrw" in name(r) or "bwd_weight" in name(r) or "BwdWeight" in name(r)]
    origin = wrw[0]["t0"] if wrw else step[0]["t0"]
    us = lambda t: (t - origin) / 1e3
    qcol = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
    print("step: %d kernels, %.2f ms from the first weight-gradient kernel to the optimizer's last kernel" % (len(step), us(step[-1]["t1"]) / 1e3))
    if first_norm is not None:
        print("backward ends (first kernel of the gradient-norm): %9.1f us" % us(first_norm))
    print("%-6s %-8s %12s %12s %10s   %s" % ("bucket", "queue", "pack start", "pack end", "us", "next RCCL kernel (start, us)"))
    for k, p in enumerate(packs):
        nxt = [r for r in step if is_rccl(r) and r["t0"] >= p["t0"]]
        nx = "%9.1f %8.1f  %s" % (us(nxt[0]["t0"]), (nxt[0]["t1"] - nxt[0]["t0"]) / 1e3, name(nxt[0])[:50]) if nxt else "(none: a one-rank group reduces nothing)"
        print("%-6d %-8s %12.1f %12.1f %10.1f   %s" % (k, p[qcol] if qcol else "-", us(p["t0"]), us(p["t1"]), (p["t1"] - p["t0"]) / 1e3, nx))
    if first_norm is not None and packs:
        print("last pack ends %.1f us %s the end of the backward pass" % (abs(us(packs[-1]["t1"]) - us(first_norm)), "after" if packs[-1]["t1"] > first_norm else "before"))
    queues = {}
    for r in step:
        queues.setdefault(r[qcol] if qcol else "-", [0, 0])
        queues[r[qcol] if qcol else "-"][0] += 1
        queues[r[qcol] if qcol else "-"][1] += r["t1"] - r["t0"]
    print("kernels / busy ms per queue:", {q: (n, round(t / 1e6, 2)) for q, (n, t) in sorted(queues.items())})


if __name__ == "__main__":
    main()
