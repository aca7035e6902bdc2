"""The DCN weight-gradient GEMM of BASELINE configs[3] ([1152, 16896] x [16896, 128] per group: a long reduction into 9 output tiles):
torch.mm against a split of the reduction into S batched pieces + a sum."""
import torch
G, N, K, Og, O = 4, 16896, 1152, 128, 512
col = torch.randn(G, N, K, device="cuda")
go2d = torch.randn(N, O, device="cuda")
out = torch.empty(G, K, Og, device="cuda")


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def plain():
    for g in range(G):
        torch.mm(col[g].t(), go2d[:, g * Og:(g + 1) * Og], out=out[g])


ref = None
plain(); ref = out.clone()
print("torch.mm x4: %.1f us" % t(plain))
for S in (4, 8, 16, 32):
    part = torch.empty(G, S, K, Og, device="cuda")

    def split():
        for g in range(G):
            a = col[g].view(S, N // S, K).transpose(1, 2)
            b = go2d.view(S, N // S, O)[:, :, g * Og:(g + 1) * Og]
            torch.bmm(a, b, out=part[g])
        torch.sum(part, 1, out=out)
    split()
    err = float((out - ref).abs().max() / ref.abs().max())
    print("split S=%d: %.1f us  (rel diff %.1e)" % (S, t(split), err))
