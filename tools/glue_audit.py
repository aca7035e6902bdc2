#!/usr/bin/env python3
"""Which lines of THIS repository launch layout / copy / cast kernels inside the training step (review item: the hot-path
glue audit).  Runs a few steps of a BASELINE config under torch.profiler (CPU + GPU activities, Python stacks), links every
GPU kernel to the ATen op that launched it and that op to the innermost mm_training_amd/ source line on its Python stack, and
prints, per (source line, kernel family), launches and GPU time per step.  Run on the GPU box:
    python tools/glue_audit.py [--config cfg4] [--steps 3] [--all]
Default: copy / cast / layout kernels only (direct_copy, elementwise copy functors, cat, fill, flip, where); --all lists every
kernel launched from a line of this repository."""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.use_shipped_miopen_db()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

GLUE = re.compile(r"copy|Copy|CatArray|cat_|FillFunctor|flip|where|to_copy|contiguous|transpose|permute", re.I)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--all", action="store_true")
    args = ap.parse_args()
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    cfg = make_config(args.config)
    torch.manual_seed(0)
    np.random.seed(0)
    ts = TrainStep(cfg, dev)
    batches = [synthetic_batch(cfg, dev, seed=i) for i in range(2)]
    for i in range(6):
        ts(batches[i % 2])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for i in range(args.steps):
            ts(batches[i % 2])
        torch.cuda.synchronize()
    events = prof.events()
    rows = collections.defaultdict(lambda: [0, 0.0])
    total = [0, 0.0]
    for ev in events:
        if not ev.kernels:
            continue
        # the innermost frame of this repository on the op's Python stack
        src = "(outside mm_training_amd: torch / autograd engine)"
        for fr in (ev.stack or []):
            m = re.search(r"(mm_training_amd/[\w/]+\.py)\((\d+)\): (\w+)", fr)
            if m:
                src = f"{m.group(1)}:{m.group(2)} {m.group(3)}"
                break
        for k in ev.kernels:
            fam = re.sub(r"<.*", "", k.name)[:60]
            if not args.all and not (GLUE.search(k.name) or GLUE.search(ev.name)):
                continue
            shapes = str([tuple(x) for x in (ev.input_shapes or []) if x])[:70]
            key = (src, ev.name[:40] + " " + shapes, fam)
            rows[key][0] += 1
            rows[key][1] += k.duration
            total[0] += 1
            total[1] += k.duration
    out = [dict(source=k[0], op=k[1], kernel=k[2], launches_per_step=v[0] / args.steps, us_per_step=v[1] / args.steps) for k, v in rows.items()]
    out.sort(key=lambda r: -r["us_per_step"])
    print(json.dumps(dict(config=args.config, steps=args.steps, launches_per_step=total[0] / args.steps, us_per_step=total[1] / args.steps,
                          rows=out[:120]), indent=1))


if __name__ == "__main__":
    main()
