import torch, time
for mb in (100, 400, 1600):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"copy {mb} MB: {us:.1f} us -> {2*mb*1.048576/us*1e3:.0f} GB/s (read+write)")
    e0.record()
    for _ in range(20): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"sum  {mb} MB: {us:.1f} us -> {mb*1.048576/us*1e3:.0f} GB/s (read)")
    e0.record()
    for _ in range(20): y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"fill {mb} MB: {us:.1f} us -> {mb*1.048576/us*1e3:.0f} GB/s (write)")
