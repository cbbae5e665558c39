import torch, time
for mb in (64, 256, 1024):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1); torch.cuda.synchronize()
    for _ in range(3): b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy %4d MB: %.1f us  read+write %.2f TB/s" % (mb, ms * 1e3, 2 * n / ms / 1e9))
    e0.record()
    for _ in range(20): a.fill_(3)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("fill %4d MB: %.1f us  write %.2f TB/s" % (mb, ms * 1e3, n / ms / 1e9))
    x = a.view(torch.int32)
    e0.record()
    for _ in range(20): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("sum  %4d MB: %.1f us  read %.2f TB/s" % (mb, ms * 1e3, n / ms / 1e9))
