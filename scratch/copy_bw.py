import torch, time
d = torch.device('cuda:0')
for mb in (37, 74, 148):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype = torch.bfloat16, device = d).normal_()
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
    s.record()
    for _ in range(20): y.copy_(x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f'{mb} MB copy: {ms*1e3:.1f} us  {2*mb*1.048576/ms:.0f} GB/s (read+write)')
    s.record()
    for _ in range(20): z = x.float().sum() if False else torch.relu_(y)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f'{mb} MB in-place relu: {ms*1e3:.1f} us  {2*mb*1.048576/ms:.0f} GB/s')
