import torch, time
x = torch.empty(537*1024*1024//4, dtype=torch.int32, device='cuda')
y = torch.empty(181*1024*1024//4, dtype=torch.int32, device='cuda')
z = torch.empty_like(x)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
print('fill 537 MiB: %.1f us' % t(lambda: x.fill_(7)))
print('copy 537 MiB -> 537 MiB: %.1f us' % t(lambda: z.copy_(x)))
print('read-reduce 181 MiB: %.1f us' % t(lambda: y.sum()))
