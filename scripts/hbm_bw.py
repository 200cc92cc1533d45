import torch, time
x = torch.empty(3 * 1024**3 // 4, device="cuda"); y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 4 / 1e9
ms = t(lambda: y.copy_(x)); print(f"copy  {2*gb/ms:.2f} TB/s (r+w) {ms:.3f} ms")
ms = t(lambda: y.fill_(1.0)); print(f"fill  {gb/ms:.2f} TB/s (w)")
ms = t(lambda: x.sum()); print(f"sum   {gb/ms:.2f} TB/s (r)")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   {2*gb/ms:.2f} TB/s (r+w)")
