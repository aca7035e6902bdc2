"""Gradient clipping + AdamW of the training step (exps/mm_training_aim.py:575-608 -- Lightning's
``gradient_clip_val=2`` and ``torch.optim.AdamW(lr=1e-3/64*bs, weight_decay=1e-7)``) as two HIP launches over every parameter
(``mmt_clip_adamw_step``, csrc/clip_adamw.hip) instead of torch's multi-tensor norm + multiply of all gradients + fused AdamW:
the multiply's pass over the gradients is folded into the update.

``ClipAdamW`` is a ``torch.optim.Optimizer`` with AdamW's parameter groups and state (``step``, ``exp_avg``, ``exp_avg_sq``: its
``state_dict`` loads into ``torch.optim.AdamW`` and back), so learning-rate schedulers and checkpoints work unchanged.  Its
``step()`` clips to ``max_norm`` (the total norm over ALL groups, like ``clip_grad_norm_`` over ``model.parameters()``) and updates;
``last_norm`` holds (total norm, clip coefficient) of the last step as a device tensor -- no host sync.  Parameters that are not
dense fp32 CUDA tensors, or ``amsgrad`` / ``maximize``, are refused: the caller keeps torch's optimizer for those.
"""
import torch

from .. import _lib

CHUNK = 65536
RING = 4          # pinned staging buffers for the gradients' addresses (see _plan)


def _dense(p):
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


def _same_layout(a, b):
    """Same memory order: equal strides wherever the size is not 1 (a 1 x 1 convolution's weight has the same bytes in either format)."""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n != 1)


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("ClipAdamW: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm)
        self.last_norm = None
        self.shadows = {}       # parameter -> bf16 copy kept current by step() (see make_bf16_shadows)
        self._plans = {}        # group index -> (key of the participating parameters, device tables)
        for group in self.param_groups:
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and _dense(p)):
                    raise TypeError("ClipAdamW takes dense fp32 CUDA parameters")

    # ---- tables of one group: which parameters take part (those with a gradient), their chunks, the static pointer arrays
    def _plan(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        # the tables hold raw device addresses: a parameter whose storage was replaced after the tables were built (module.to(),
        # `p.data = ..`, a memory-format conversion) must not be written through the old one -- its address is part of the key
        # (the moments only change through load_state_dict, which drops the tables)
        key = tuple((id(p), p.data_ptr()) for p in ps)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        dev = ps[0].device
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # (moments follow the parameter's own memory order: a channels_last weight and its moments line up element by element)
            if not (_same_layout(st["exp_avg"], p) and _same_layout(st["exp_avg_sq"], p)):
                raise RuntimeError("ClipAdamW: a moment's layout differs from its parameter's")
        # One launch de-biases every tensor of the group with the SAME step count (torch.optim.AdamW keeps one per parameter).  The
        # counts only diverge when the participating set changes -- a parameter that starts receiving gradients later, a loaded
        # state with differing counts -- and every such change rebuilds these tables: refuse it here, loudly, instead of
        # de-biasing the newcomer with the wrong power of beta.
        counts = {int(self.state[p]["step"]) for p in ps}
        if len(counts) > 1:
            raise RuntimeError(f"ClipAdamW: the parameters of group {gi} that have gradients carry different step counts {sorted(counts)[:4]}: "
                               "per-parameter bias correction is not supported (a parameter that joins later belongs in a parameter group of its own)")
        chunk_t, chunk_o = [], []
        for t, p in enumerate(ps):
            for off in range(0, p.numel(), CHUNK):
                chunk_t.append(t)
                chunk_o.append(off)
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        plan = dict(key=key, params=ps, n_chunks=len(chunk_t),
                    chunk_t=torch.tensor(chunk_t, dtype=torch.int32, device=dev), chunk_o=i64(chunk_o),
                    p=i64([p.data_ptr() for p in ps]), m=i64([self.state[p]["exp_avg"].data_ptr() for p in ps]),
                    v=i64([self.state[p]["exp_avg_sq"].data_ptr() for p in ps]), numel=i64([p.numel() for p in ps]),
                    shadow=i64([self.shadows[p].data_ptr() if p in self.shadows else 0 for p in ps]) if self.shadows else None,
                    # the gradients' addresses change from step to step (autograd allocates them afresh): staged through a RING of pinned
                    # buffers, each guarded by an event -- the host runs ahead of the device by more than a step, and a single buffer
                    # would be overwritten with the next step's addresses before this step's copy has executed
                    g_host=[torch.empty(len(ps), dtype=torch.int64).pin_memory() for _ in range(RING)], g_event=[None] * RING, g_turn=0,
                    g=torch.empty(len(ps), dtype=torch.int64, device=dev),
                    partials=torch.empty(max(len(chunk_t), 1), dtype=torch.float32, device=dev),
                    norm=torch.zeros(2, dtype=torch.float32, device=dev))
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def make_bf16_shadows(self, params):
        """A bf16 copy of every given parameter (same shape and memory order), rewritten by every step() with the updated values:
        what an autocast convolution would otherwise produce with a cast kernel per layer and step (ops/conv_overlap.py looks the
        copy up and remembers the parameter's version counter: step() updates the parameter through raw pointers and leaves the counter
        alone, so any OTHER in-place change of the parameter (load_state_dict, an initialiser) shows as a mismatch and the consumer
        refreshes the copy)."""
        from ..ops import conv_overlap
        for p in params:
            sh = torch.empty_like(p, dtype=torch.bfloat16, memory_format=torch.preserve_format)
            sh.copy_(p)
            self.shadows[p] = sh
            conv_overlap.register_bf16_shadow(p, sh)
        self._plans.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(gi, g) for gi, g in enumerate(self.param_groups) if any(p.grad is not None for p in g["params"])]
        if self.max_norm > 0.0 and len(groups) > 1:
            raise RuntimeError("ClipAdamW: clipping over several parameter groups is not supported (one group holds the model)")
        for gi, group in groups:
            plan = self._plan(gi, group)
            ps = plan["params"]
            turn = plan["g_turn"]
            plan["g_turn"] = (turn + 1) % RING
            if plan["g_event"][turn] is not None:
                plan["g_event"][turn].synchronize()                      # (the copy that last read this buffer: RING steps ago)
            addrs = []
            for p in ps:
                g = p.grad
                if g.dtype != torch.float32 or not _same_layout(g, p):       # (not met in the step: autograd hands gradients over in the parameter's layout)
                    g = p.grad = torch.empty_like(p).copy_(g)
                addrs.append(g.data_ptr())
            host = plan["g_host"][turn]
            host.copy_(torch.tensor(addrs, dtype=torch.int64))
            plan["g"].copy_(host, non_blocking=True)
            ev = plan["g_event"][turn] = plan["g_event"][turn] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(ps[0].device))
            st0 = self.state[ps[0]]
            step = int(st0["step"]) + 1
            for p in ps:
                self.state[p]["step"] += 1
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                _lib.call("mmt_clip_adamw_step", plan["n_chunks"], CHUNK, plan["chunk_t"].data_ptr(), plan["chunk_o"].data_ptr(),
                          plan["p"].data_ptr(), plan["g"].data_ptr(), plan["m"].data_ptr(), plan["v"].data_ptr(),
                          plan["shadow"].data_ptr() if plan["shadow"] is not None else 0, plan["numel"].data_ptr(),
                          float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                          float(self.max_norm), plan["partials"].data_ptr(), plan["norm"].data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
            self.last_norm = plan["norm"]
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()                       # (the moments are new tensors)
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]) and st["step"].is_cuda:
                st["step"] = st["step"].cpu()     # (torch's fused AdamW keeps the count on the device; here it is a host value)
