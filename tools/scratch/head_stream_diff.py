"""Where do the two-stream and single-stream gradients of the task heads differ (step 5 of tests/test_head_streams_gpu.py)?"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.dp import make_config
from mm_training_amd.layers.heads.bev_depth_head import BEVDepthHead

cfg = make_config("tiny")
torch.manual_seed(0)
one = BEVDepthHead(**cfg["head_conf"]).cuda()
for m in one.modules():
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        m.to(memory_format=torch.channels_last)
two = copy.deepcopy(one)


def run(head, x, streams):
    head.zero_grad(set_to_none=True)
    xi = x.clone().requires_grad_(True)
    out = head._forward_tasks_on_streams(xi, streams) if streams else tuple([task(xi)] for task in head.task_heads)
    loss = sum((v.float() * (1 + i)).square().mean() for i, task in enumerate(out) for v in task[0].values())
    loss.backward()
    torch.cuda.synchronize()
    return xi.grad.clone(), out


for step in range(8):
    x = torch.randn(2, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    res = [run(one, x, 0), run(one, x, 0), run(two, x, 2), run(two, x, 2)]
    ga = res[0][0]
    line = []
    for name, (g, _) in zip(("a2", "b", "b2"), res[1:]):
        d = (g - ga).abs()
        big = d > 1e-4 * ga.abs().max()
        idx = big.nonzero()
        ext = "" if idx.numel() == 0 else " n%s y[%d..%d] x[%d..%d]" % (sorted(set(idx[:, 0].tolist())), idx[:, 2].min(), idx[:, 2].max(), idx[:, 3].min(), idx[:, 3].max())
        line.append("%s: max %.2e, %d elements above 1e-4%s" % (name, float(d.max() / ga.abs().max()), int(big.sum()), ext))
    print("step", step, " | ".join(line), flush=True)
