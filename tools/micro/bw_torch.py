import torch
x=torch.empty(3300*1024*1024//2, dtype=torch.bfloat16, device='cuda')
y=torch.empty(3300*1024*1024//8, dtype=torch.bfloat16, device='cuda')
def t(f,n=5):
    f(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(True),torch.cuda.Event(True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
ms=t(lambda: x.zero_()); print('fill 3.3GB', ms, 'ms', 3.46/ms, 'TB/s')
z=torch.empty_like(x)
ms=t(lambda: z.copy_(x)); print('copy 3.3GB', ms, 'ms', 2*3.46/ms, 'TB/s')
ms=t(lambda: y.sum()); print('read 0.83GB', ms)
w=torch.empty(3300*1024*1024//2//4,4, dtype=torch.bfloat16, device='cuda')
ms=t(lambda: torch.add(y.view(-1,1), 0, out=None) ); print('r+w small',ms)
