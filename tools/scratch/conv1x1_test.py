import torch, time
torch.backends.cudnn.benchmark = False
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
shapes = [(24,64,176,64,256),(24,64,176,256,64),(24,32,88,512,128),(24,32,88,128,512),(24,16,44,1024,256),(24,16,44,256,1024),(24,8,22,2048,512),(24,8,22,512,2048),(24,16,44,512,512)]
for (N,H,W,Ci,Co) in shapes:
    x = torch.randn(N,Ci,H,W,device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv = torch.nn.Conv2d(Ci,Co,1,bias=False).cuda().to(memory_format=torch.channels_last)
    w = conv.weight.detach().reshape(Co,Ci).clone().requires_grad_(True)
    def f_conv():
        y = conv(x); y.backward(torch.ones_like(y)) ; return y
    def f_mm():
        xm = x.permute(0,2,3,1).reshape(-1,Ci)
        y = xm @ w.t(); y.backward(torch.ones_like(y)); return y
    flops = 3*2*N*H*W*Ci*Co
    tc, tm = t(f_conv), t(f_mm)
    print(f"N{N} {H}x{W} {Ci}->{Co}: conv {tc:.3f} ms ({flops/tc/1e9:.1f} TF)  matmul {tm:.3f} ms ({flops/tm/1e9:.1f} TF)")
# 3x3
for (N,H,W,Ci,Co) in [(24,64,176,64,64),(24,32,88,128,128),(24,16,44,256,256),(24,16,44,512,512),(4,32,32,160,160)]:
    x = torch.randn(N,Ci,H,W,device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv = torch.nn.Conv2d(Ci,Co,3,padding=1,bias=False).cuda().to(memory_format=torch.channels_last)
    def f_conv():
        y = conv(x); y.backward(torch.ones_like(y)); return y
    flops = 3*2*N*H*W*Ci*Co*9
    tc = t(f_conv)
    print(f"3x3 N{N} {H}x{W} {Ci}->{Co}: conv {tc:.3f} ms ({flops/tc/1e9:.1f} TF)")
