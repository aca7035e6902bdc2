"""GPU, world_size 2 (two processes sharing cuda:0, gloo rendezvous on 127.0.0.1): the data-parallel
step of the full tiny model through the HIP kernels.  SURVEY section 8e correctness check: two ranks
with batch b each reproduce the detection-loss gradient of one process with batch 2b (eval-mode
BatchNorm / dropout so that samples do not interact; the depth loss is normalised per rank in the
reference and is left out)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice(batch, lo, hi):
    imgs, mats, pcs, boxes, labels = batch
    m = {k: (v[lo * (v.shape[0] // imgs.shape[0]):hi * (v.shape[0] // imgs.shape[0])] if torch.is_tensor(v) else v)
         for k, v in mats.items()}
    return imgs[lo:hi], m, pcs[lo:hi], boxes[lo:hi], labels[lo:hi]


def _grads(ts, batch):
    ts.model.eval()
    ts.optimizer.zero_grad(set_to_none=True)
    _, det, _ = ts.forward_loss(batch)
    det.backward()
    return {n: p.grad.detach().float().cpu().clone() for n, p in ts.model.named_parameters() if p.grad is not None}, float(det.detach())


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = make_config("tiny")
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev, world_size=world)
    full = synthetic_batch(cfg, dev, seed=7, batch_size=2 * world)
    grads, det = _grads(ts, _slice(full, 2 * rank, 2 * rank + 2))
    if rank == 0:
        out["grads"], out["det"] = grads, det
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process_double_batch(mmt_lib):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    dev = torch.device("cuda", 0)
    cfg = make_config("tiny")
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev, world_size=1)
    ref, _ = _grads(ts, synthetic_batch(cfg, dev, seed=7, batch_size=4))
    got = out["grads"]
    assert set(got) == set(ref) and len(ref) > 50
    rel = {}
    for n in ref:
        rel[n] = ((got[n] - ref[n]).norm() / ref[n].norm().clamp(min=1e-12)).item()
    top = sorted(rel.items(), key=lambda kv: -kv[1])[:6]
    vals = sorted(rel.values())
    # A data-parallel bug (missing all-reduce, wrong loss normaliser, wrong shard) gives O(1) errors.  What is
    # tolerated here: fp32 atomics of the pooling / DCN kernels, and MIOpen picking different convolution
    # solvers in different processes on a box with a cold kernel cache (seen: 3 % on one head weight).
    assert vals[len(vals) // 2] <= 2e-3 and top[0][1] <= 8e-2, top
