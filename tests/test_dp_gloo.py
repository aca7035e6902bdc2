"""CPU, world_size 2, gloo: the data-parallel path.  The hot-path kernels are per-sample
(no data-path collective); what DP adds is (1) sharding the batch,
(2) the gradient all-reduce (mean) and (3) the cross-rank loss normalisers of the head
(bev_depth_head.py:273-276,300-301).  Checked on the dense part of the model, which runs
on CPU: two ranks with batch b each == one process with batch 2b."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    from mm_training_amd.dp.configs import make_config
    from mm_training_amd.layers.heads.bev_depth_head import BEVDepthHead
    cfg = make_config("tiny")
    torch.manual_seed(seed)
    head = BEVDepthHead(**cfg["head_conf"])
    # BatchNorm statistics are per-rank in the reference (no SyncBN): use eval-mode BN so the
    # 2-rank and 1-rank runs are comparable sample by sample
    head.eval()
    return cfg, head


def _data(cfg, n, seed=1, every_class=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cfg["fuse_layer_in_channels"], 128, 128, generator=g)
    boxes, labels = [], []
    for i in range(n):
        k = 8 if every_class else 3 + i
        xy = torch.rand(k, 2, generator=g) * 80 - 40
        rest = torch.tensor([[-1.0, 1.9, 4.6, 1.7, 0.3, 0.5, -0.2]]).repeat(k, 1)
        boxes.append(torch.cat([xy, rest], 1))
        # every_class: two boxes of each of the 4 classes in EVERY sample, so that no rank's per-task positive count is 0 -- the
        # reference clamps the cross-rank MEAN of the counts at 1 (bev_depth_head.py:273-276), which equals the single-process
        # normaliser max(sum, 1) / world only while that mean is >= 1
        labels.append(torch.arange(k) % 4 if every_class else torch.randint(0, 4, (k,), generator=g))
    return x, boxes, labels


def _step(head, x, boxes, labels):
    preds = head(x)
    targets = head.get_targets_torch(boxes, labels)   # torch restatement: the HIP op needs a GPU
    return head.loss(targets, preds)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg, head = _make()
    ddp = torch.nn.parallel.DistributedDataParallel(head, find_unused_parameters=True)
    x, boxes, labels = _data(cfg, 4)
    sl = slice(rank * 2, rank * 2 + 2)
    preds = ddp(x[sl])
    loss = head.loss(head.get_targets_torch(boxes[sl], labels[sl]), preds)
    loss.backward()
    if rank == 0:
        out["grads"] = {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}
        out["loss"] = loss.item()
    dist.destroy_process_group()


def test_two_rank_ddp_equals_single_process_double_batch():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    cfg, head = _make()
    x, boxes, labels = _data(cfg, 4)
    # single process, global batch 4: the loss normalisers are global sums there, while each
    # DP rank divides by the cross-rank MEAN of its normalisers (reduce_mean): grads then
    # average to (1/world) * sum_r grad(L_r) with L_r = S_r / mean(N) => 1-proc gradient.
    loss = _step(head, x, boxes, labels)
    loss.backward()
    ref = {n: p.grad for n, p in head.named_parameters() if p.grad is not None}
    got = out["grads"]
    assert set(got) == set(ref)
    for n in ref:
        assert torch.allclose(got[n], ref[n], rtol=2e-3, atol=2e-5), n


def test_loss_normaliser_is_one_collective_and_device_resident():
    """No .item(): the loss is a tensor with grad_fn and needs no process group when world=1."""
    cfg, head = _make()
    x, boxes, labels = _data(cfg, 2)
    loss = _step(head, x, boxes, labels)
    assert loss.requires_grad and loss.dim() == 0


def _reducer_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from mm_training_amd.dp.reducer import GradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1), torch.nn.Flatten(),
                              torch.nn.Linear(4 * 8 * 8, 5))
    net[0].to(memory_format=torch.channels_last)
    frozen = torch.nn.Linear(3, 3)                         # never used: must be kept out of the reducer
    red = GradReducer(list(net.named_parameters()) + [("frozen." + n, p) for n, p in frozen.named_parameters()], world,
                      bucket_mb=0.002, first_mb=0.0004, last_mb=0.0005, ignore=("frozen.",))   # 2 KB buckets, a smaller first and last one
    g = torch.Generator().manual_seed(100)
    data = torch.randn(world * 4, 3, 8, 8, generator=g)
    res = {}
    for step in range(2):
        net.zero_grad(set_to_none=True)
        net(data[rank * 4:(rank + 1) * 4]).square().mean().backward()
        red.finish()
        res[step] = [p.grad.clone() for p in net.parameters()]
    # a backward pass that dies half-way must not poison the next one
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, gr):
            raise RuntimeError("boom")
    net.zero_grad(set_to_none=True)
    try:
        net[4](net[3](Boom.apply(net[2](net[1](net[0](data[:4])))))).sum().backward()
    except RuntimeError:
        pass
    net.zero_grad(set_to_none=True)
    net(data[rank * 4:(rank + 1) * 4]).square().mean().backward()
    red.finish()
    res["after_failure"] = [p.grad.clone() for p in net.parameters()]
    # a layer applied twice in one graph (one weight, two uses): its hook fires once, with the sum
    net.zero_grad(set_to_none=True)
    red.begin_step()
    shard = data[rank * 4:(rank + 1) * 4]
    h = net[1](net[0](shard))
    y = net[2](h) + net[2](h.flip(-1))
    net[4](net[3](y)).square().mean().backward()
    red.finish()
    res["shared"] = [p.grad.clone() for p in net.parameters()]
    if rank == 0:
        out["grads"] = res
        out["describe"] = red.describe()
        out["strides_match"] = all(p.grad.stride() == p.stride() for p in net.parameters())
        out["data"] = data
    dist.barrier()
    dist.destroy_process_group()


def test_native_reducer_two_ranks_equal_the_mean_gradient():
    """dp/reducer.py on CPU tensors over gloo: two ranks with a shard each end up with the gradient of the mean of the two shard
    losses = the single-process gradient of (loss(shard 0) + loss(shard 1)) / 2; several buckets, channels_last weights keep
    their layout, an unused parameter is ignored, a failed backward pass does not poison the next."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reducer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1), torch.nn.Flatten(),
                              torch.nn.Linear(4 * 8 * 8, 5))
    data = out["data"]
    loss = (net(data[:4]).square().mean() + net(data[4:]).square().mean()) / 2
    loss.backward()
    ref = [p.grad for p in net.parameters()]
    for key in (0, 1, "after_failure"):
        for a, b in zip(out["grads"][key], ref):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), key
    net.zero_grad(set_to_none=True)
    def twice(x):
        h = net[1](net[0](x))
        return net[4](net[3](net[2](h) + net[2](h.flip(-1)))).square().mean()
    ((twice(data[:4]) + twice(data[4:])) / 2).backward()
    for a, b in zip(out["grads"]["shared"], [p.grad for p in net.parameters()]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), "shared"
    d = out["describe"]
    # not uniform: the first and the last bucket are the small ones
    bb = d["bucket_bytes"]
    assert bb[0] <= 0.0004 * 2 ** 20 + 1 or len(bb) < 3
    assert bb[-1] == 864                       # the first layer's weight alone (a parameter larger than the cap is a bucket of its own): 8 x 3 x 3 x 3 floats
    assert d["buckets"] >= 3 and d["parameters"] == 6 and sum(d["bucket_bytes"]) == d["gradient_bytes"] and out["strides_match"]


class _HeadWithUnused(torch.nn.Module):
    """The tiny config's detection head plus a `context_se` block whose parameters never receive a gradient -- what DepthNet
    carries (lss_fpn.py:183) and what dp/trainer.py tells the reducer to ignore by that name."""

    def __init__(self, head):
        super().__init__()
        self.depth_net = torch.nn.Module()
        self.depth_net.context_se = torch.nn.Linear(4, 4)
        self.head = head

    def forward(self, x):
        return self.head(x)


def _world4_worker(rank, world, port, out):
    from mm_training_amd.dp.reducer import GradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg, head = _make()
    model = _HeadWithUnused(head)
    red = GradReducer(model.named_parameters(), world, bucket_mb=0.5, first_mb=0.02, last_mb=0.05, ignore=(".context_se.",))
    opt = torch.optim.SGD([p for p in model.parameters()], lr=1e-5, momentum=0.9)
    for step in range(3):
        x, boxes, labels = _data(cfg, world, seed=10 + step, every_class=True)
        opt.zero_grad(set_to_none=True)
        sl = slice(rank, rank + 1)
        loss = head.loss(head.get_targets_torch(boxes[sl], labels[sl]), model(x[sl]))
        loss.backward()
        red.finish()
        opt.step()
    if rank == 0:
        out["params"] = {n: p.detach().clone() for n, p in model.named_parameters()}
        out["describe"] = red.describe()
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_native_reducer_three_steps_equal_one_process_with_the_whole_batch():
    """World 4 over gloo, the tiny config's head through dp/reducer.py (not torch's DDP): after three optimiser steps every
    parameter equals the single-process run on the 4-sample batch -- the gradient is the MEAN over the ranks and the head's loss
    normalisers are global sums turned into means by one collective (bev_depth_head.py:273-276, :300-301) -- with `.context_se.`
    parameters kept out of the reducer (they never get a gradient: an incomplete bucket would raise) and at least three buckets
    of different sizes (small first and last one)."""
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world4_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    cfg, head = _make()
    model = _HeadWithUnused(head)
    opt = torch.optim.SGD([p for p in model.parameters()], lr=1e-5, momentum=0.9)
    for step in range(3):
        x, boxes, labels = _data(cfg, world, seed=10 + step, every_class=True)
        opt.zero_grad(set_to_none=True)
        _step(head, x, boxes, labels).backward()
        opt.step()
    got, d = out["params"], out["describe"]
    init = dict(_HeadWithUnused(_make()[1]).named_parameters())
    moved = 0
    for n, p in model.named_parameters():
        # the UPDATES agree (the weights are O(1), three small steps move them by far less: compare what moved)
        du, dr = got[n] - init[n].detach(), p.detach() - init[n].detach()
        ulp = 1.2e-7 * float(init[n].detach().abs().max())            # the weights themselves round to fp32: a few ulps of slack
        assert torch.isfinite(dr).all() and torch.allclose(du, dr, rtol=1e-3, atol=1e-3 * float(dr.abs().max()) + 4 * ulp + 1e-12), n
        moved += int(float(dr.abs().max()) > 0)
        if ".context_se." in n:
            assert float(du.abs().max()) == 0.0
    assert moved > 10                                       # (the steps did change the weights)
    bb = d["bucket_bytes"]
    assert d["buckets"] >= 3 and len(set(bb)) >= 2 and bb[0] < max(bb) and bb[-1] < max(bb), d
    assert d["parameters"] == sum(1 for n, p in model.named_parameters() if ".context_se." not in n and p.requires_grad)
