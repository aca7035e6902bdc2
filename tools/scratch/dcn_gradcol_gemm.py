"""The DCN column-gradient GEMM of BASELINE configs[3] (per group [16896, 128] x [128, 1152] -> 78 MB of output: write-bound):
four torch.mm against one strided-batch bmm, and against the transposed product."""
import torch
G, N, K, Og, O = 4, 16896, 1152, 128, 512
wmat = torch.randn(G, K, Og, device="cuda")
go2d = torch.randn(N, O, device="cuda")
out = torch.empty(G, N, K, device="cuda")


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
        torch.mm(go2d[:, g * Og:(g + 1) * Og], wmat[g].t(), out=out[g])


def batched():
    torch.bmm(go2d.view(N, G, Og).permute(1, 0, 2), wmat.transpose(1, 2), out=out)


plain(); ref = out.clone()
print("torch.mm x4: %.1f us" % t(plain))
batched(); print("one bmm: %.1f us (max diff %.1e)" % (t(batched), float((out - ref).abs().max())))
for S in (2, 4, 8):
    def rows():
        for g in range(G):
            a = go2d.view(S, N // S, O)[:, :, g * Og:(g + 1) * Og]
            torch.bmm(a, wmat[g].t().unsqueeze(0).expand(S, Og, K), out=out[g].view(S, N // S, K))
    rows(); print("rows cut into %d: %.1f us (max diff %.1e)" % (S, t(rows), float((out - ref).abs().max())))
