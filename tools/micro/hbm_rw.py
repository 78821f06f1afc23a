import torch, time
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
n=2*80*160*160*32
a=torch.empty(n,dtype=torch.bfloat16,device='cuda'); b=torch.empty_like(a); c=torch.empty_like(a)
print("fill 262 MB: %.1f us"%t(lambda: a.fill_(1.0)))
print("copy 262->262 MB: %.1f us"%t(lambda: b.copy_(a)))
print("add 2x262 -> 262: %.1f us"%t(lambda: torch.add(a,b,out=c)))
print("sum 262 MB read: %.1f us"%t(lambda: a.float().sum() if False else torch.sum(a.view(torch.int16)[::1][:n//2].to(torch.int32)) if False else a.view(torch.int32).sum()))
