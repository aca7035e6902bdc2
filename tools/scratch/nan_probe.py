import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.dp import make_config, TrainStep, synthetic_batch
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
cfg = make_config(name); cfg["dtype"] = "bf16"
dev = torch.device("cuda", 0)
ts = TrainStep(cfg, dev)
bad = []
def hook(mod_name):
    def f(m, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        for o in outs:
            if torch.is_tensor(o) and o.is_floating_point() and not torch.isfinite(o).all():
                if len(bad) < 5:
                    bad.append(mod_name); print("NONFINITE after", mod_name, type(m).__name__, o.dtype, tuple(o.shape), flush=True)
    return f
for n, m in ts.model.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_hook(hook(n))
batch = synthetic_batch(cfg, dev)
loss, det, dep = ts.forward_loss(batch)
print("loss", float(loss), float(det), float(dep))
