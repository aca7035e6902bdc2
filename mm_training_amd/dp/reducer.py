"""Bucketed gradient all-reduce for the data-parallel step, overlapped with the backward pass on a side HIP stream.

What torch's DistributedDataParallel does for the reference (Lightning's DDP strategy, exps/mm_training_aim.py:595-612), rebuilt for
this step's static graph because DDP's per-PARAMETER work is what a rank pays before any communication: its reducer copies (and
scales) every gradient into its bucket with one small kernel per parameter -- ~400 launches and 3.2 ms per step at BASELINE
configs[3] on ONE rank (tools/scratch/ddp_tax.py: 70.5 ms against 67.3 without the wrap) -- and reads each gradient inside the
backward pass on the main stream, which rules out the deferred weight-gradient stream of ops/conv_overlap.py.

Here:
  * the parameters that receive a gradient are cut, in reverse registration order (the order the backward pass produces them),
    into buckets of `bucket_mb`; each bucket owns one flat buffer and a view per parameter;
  * a post-accumulate-grad hook per parameter counts its bucket down; when a bucket is complete, the COMMUNICATION stream waits
    for every stream of the step that may still be producing one of its gradients (main, the task heads' streams, the
    weight-gradient stream -- whatever has been queued there by now includes the producers, the hooks fire behind them in
    host order), packs the gradients with ONE multi-tensor copy, scales the flat buffer once and starts the all-reduce
    (RCCL; async) -- so per step there are a few launches per BUCKET instead of one per parameter, none of them on the main
    stream, and nothing the main stream waits for before the end of the backward pass;
  * `finish()` (after `backward()`): the calling stream waits for the collectives and every `.grad` is re-pointed at its view of
    the reduced buffer (what `gradient_as_bucket_view` gives under DDP).
Buckets are not uniform (`cut_buckets`): a small first one (the first collective starts early), a small last one (the only
all-reduce nothing overlaps is the last: DESIGN section 6 prices it), `bucket_mb` between.
The autograd engine runs the nodes of a static graph in the same order on every rank, so the buckets complete -- and the
collectives are issued -- in the same order everywhere (the property DDP relies on too).  Works on CPU tensors (gloo) without
streams: tests/test_dp_gloo.py.  Gradient accumulation over several backward passes between two `finish()` calls is not supported
(the step never does it); a parameter that gets no gradient must not be registered (`ignore`).
"""
import torch
import torch.distributed as dist


def cut_buckets(sizes, bucket_bytes, first_bytes, last_bytes):
    """Bucket boundaries over `sizes` (bytes per parameter, in the order the backward pass produces the gradients): a list of
    (begin, end).  Not uniform: the FIRST bucket is small, so the first collective starts early in the backward pass; the LAST
    one -- the first layers' gradients, complete only when the backward pass ends -- is small too, because its all-reduce is
    the one nothing overlaps: what the step waits for after backward() is (last bucket) / (link bandwidth), not
    (bucket_mb) / (link bandwidth).  Between them `bucket_bytes` (few, large collectives: xGMI is point-to-point, a ring
    all-reduce is bound by one link per direction and every collective pays its latency once)."""
    n = len(sizes)
    tail, acc = n, 0                     # the parameters of the last bucket, from the end
    while tail > 1 and acc + sizes[tail - 1] <= last_bytes:
        tail -= 1
        acc += sizes[tail]
    if tail == n:
        tail = n - 1 if n > 1 else n     # (one huge last parameter: its own bucket)
    cuts, begin, acc, cap = [], 0, 0, first_bytes
    for i in range(tail):
        if i > begin and acc + sizes[i] > cap:
            cuts.append((begin, i))
            begin, acc, cap = i, 0, bucket_bytes
        acc += sizes[i]
    if tail > begin:
        cuts.append((begin, tail))
    if tail < n:
        cuts.append((tail, n))
    return cuts


class GradReducer:
    def __init__(self, named_parameters, world_size, bucket_mb=64, ignore=(), extra_streams=lambda: (), stream=None, first_mb=8, last_mb=8):
        # the mean is over the ranks that exchange: without a process group nothing is exchanged and nothing is scaled
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.extra_streams = extra_streams                   # callable -> streams (besides the current one) gradients may come from
        params = [(n, p) for n, p in named_parameters if p.requires_grad and not any(tag in n for tag in ignore)]
        if not params:
            raise ValueError("GradReducer: no parameters")
        if len({id(p) for _, p in params}) != len(params):
            raise ValueError("GradReducer: a parameter is registered twice (tied weights: pass named_parameters(), which lists them once)")
        self.device = params[0][1].device
        self.cuda = self.device.type == "cuda"
        # the stream the buckets are packed on (and the collective is issued from): its own, at normal priority, unless the caller
        # hands one in
        self.comm = (stream if stream is not None else torch.cuda.Stream(device=self.device)) if self.cuda else None
        mb = 1024 * 1024
        order = list(reversed(params))                       # backward produces the last layers' gradients first
        self.buckets = []
        begin = 0                                            # (a change of dtype starts a new run of buckets)
        for i in range(1, len(order) + 1):
            if i == len(order) or order[i][1].dtype != order[begin][1].dtype:
                run = order[begin:i]
                sizes = [p.numel() * p.element_size() for _, p in run]
                first = first_mb if begin == 0 else bucket_mb
                last = last_mb if i == len(order) else bucket_mb
                for lo, hi in cut_buckets(sizes, int(bucket_mb * mb), int(min(first, bucket_mb) * mb), int(min(last, bucket_mb) * mb)):
                    self.buckets.append(run[lo:hi])
                begin = i
        self._flat, self._views, self._bucket_of, self._pending, self._works = [], [], {}, [], []
        self._step, self._seen_step, self._arrived = 0, 0, set()
        for b, members in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for _, p in members), dtype=members[0][1].dtype, device=self.device)
            views, off = [], 0
            for _, p in members:
                # the view has the PARAMETER's strides (channels_last weights): autograd hands gradients over in that layout, the
                # multi-tensor copy then takes its fast path, and the fused optimizer requires param / grad layouts to agree
                dense = p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.is_contiguous()
                views.append(torch.as_strided(flat, p.size(), p.stride(), off) if dense else flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self._bucket_of[p] = b
            self._flat.append(flat)
            self._views.append(views)
            self._pending.append(len(members))
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for members in self.buckets for _, p in members]
        self.gradient_bytes = int(sum(f.numel() * f.element_size() for f in self._flat))

    def begin_step(self):
        """Called by the step before its backward pass (TrainStep does): a new pass starts from clean counters whatever became
        of the previous one."""
        self._step += 1

    # ---- one call per parameter and backward pass, on the autograd thread, right after its .grad was set
    def _on_grad(self, p):
        # a new backward pass: the step said so (begin_step), or -- for a caller that does not -- a gradient arrives that this pass
        # has seen already (AccumulateGrad runs once per parameter and pass, tied weights included: their uses are summed in front of it)
        if self._seen_step != self._step or id(p) in self._arrived:
            if any(n != len(m) for n, m in zip(self._pending, self.buckets)):
                self._reset()                                # the previous pass died half-way
            self._seen_step = self._step
            self._arrived.clear()
        self._arrived.add(id(p))
        b = self._bucket_of[p]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        members, flat, views = self.buckets[b], self._flat[b], self._views[b]
        grads = [p.grad for _, p in members]
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            self.comm.wait_stream(cur)
            for s in self.extra_streams():
                if s is not None and s != cur and s != self.comm:
                    self.comm.wait_stream(s)
            with torch.cuda.stream(self.comm):
                torch._foreach_copy_(views, grads)           # one multi-tensor launch per bucket
                if self.world > 1:
                    flat.mul_(1.0 / self.world)
                work = dist.all_reduce(flat, async_op=True) if dist.is_initialized() else None
            for g in grads:                                  # their blocks stay out of the allocator's hands until the copy has run
                g.record_stream(self.comm)
        else:
            torch._foreach_copy_(views, grads)
            if self.world > 1:
                flat.mul_(1.0 / self.world)
            work = dist.all_reduce(flat, async_op=True) if dist.is_initialized() else None
        self._works.append(work)

    def finish(self):
        """After backward(): wait for the collectives (the calling stream does, not the host) and hand every parameter its view of
        the reduced buffer as .grad."""
        missing = [n for b, members in enumerate(self.buckets) if self._pending[b] != 0 for n, p in members if p.grad is None]
        if any(self._pending):
            self._reset()
            raise RuntimeError("GradReducer: the backward pass left buckets incomplete; parameters without a gradient: %s" % missing[:6])
        for w in self._works:
            if w is not None:
                w.wait()
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.comm)
        for members, views in zip(self.buckets, self._views):
            for (_, p), v in zip(members, views):
                p.grad = v
        self._reset()

    def _reset(self):
        for w in self._works:                                # (collectives of a pass that died: every rank issued the same ones; let them end)
            if w is not None:
                try:
                    w.wait()
                except Exception:
                    pass
        self._works = []
        self._arrived.clear()
        self._pending = [len(m) for m in self.buckets]

    def describe(self):
        return {"kind": "native bucketed all-reduce (dp/reducer.py)", "buckets": len(self.buckets),
                "bucket_bytes": [int(f.numel() * f.element_size()) for f in self._flat], "gradient_bytes": self.gradient_bytes,
                "bucket_layout": "small first bucket (the first collective starts early), small last bucket (the one all-reduce nothing overlaps), bucket_mb between",
                "parameters": int(sum(len(m) for m in self.buckets)), "communication_stream": bool(self.cuda),
                "launches_per_bucket": "one multi-tensor copy + one scale + the collective"}

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
